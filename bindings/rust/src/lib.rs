//! mp3rgain_amd -- Rust binding of `libmp3rgain_amd.so`, the MI355X-native ReplayGain 1.0 analysis path.
//!
//! **This file cannot be compiled where it is developed** (the build image has neither `rustc` nor `cargo`).  It is kept
//! honest by `tests/test_rust_binding_layout.py`, which parses this file and the C headers and compares every `extern "C"`
//! declaration (name, arity, argument and return types) and every `#[repr(C)]` struct (field order, names, widths,
//! offsets, size -- against a C program compiled from the headers).
//!
//! Two layers:
//! * [`ffi`]: the C ABI of `include/mp3rgain_amd.h` and `include/mp3rgain_amd_node.h`, nothing else.
//! * [`replaygain`]: the names and signatures of mp3rgain's `src/replaygain.rs` -- `analyze_track[_with_index]`
//!   (`:929-941`), `analyze_album[_with_index]` (`:1033-1074`), `find_peak_amplitude` (`:1140`), `is_available`
//!   (`:1119`), the result structs (`:57-95`, `:1125-1132`) -- implemented over the library: a maintainer replaces
//!   `mod replaygain;` by `pub use mp3rgain_amd::replaygain;` and the callers in `src/main.rs` / `src/lib.rs` compile
//!   unchanged.  MPEG Layer III is decoded by the library (on the GPU); an M4A/AAC file needs
//!   [`replaygain::set_decoder_command`].
//!
//! The seam *after* the decoder (the host keeps symphonia and hands decoded PCM over) is
//! [`replaygain::analyze_tracks_pcm`] / [`replaygain::analyze_album_pcm`].

#![allow(non_snake_case)]

pub mod ffi {
    //! `extern "C"` declarations; one line per prototype of the headers, in the headers' order.
    use std::os::raw::{c_char, c_int, c_void};

    pub const RG_ABI_VERSION: c_int = 5;
    pub const RG_HISTOGRAM_SIZE: usize = 12000;
    pub const RG_OK: c_int = 0;
    pub const RG_ERR_INVALID_ARG: c_int = -1;
    pub const RG_ERR_UNSUPPORTED_RATE: c_int = -2;
    pub const RG_ERR_DEVICE: c_int = -3;
    pub const RG_ERR_NO_DEVICE: c_int = -4;
    pub const RG_ERR_NOMEM: c_int = -5;
    pub const RG_ERR_STATE: c_int = -6;
    pub const RG_ERR_COLLECTIVE: c_int = -7;
    pub const RG_ERR_IO: c_int = -8;
    pub const RG_ERR_FORMAT: c_int = -9;
    pub const RG_ERR_REFUSED: c_int = -10;
    pub const RG_FMT_F32_PLANAR: u16 = 0;
    pub const RG_FMT_S16_PLANAR: u16 = 1;
    pub const RG_FMT_S32_PLANAR: u16 = 2;
    pub const RG_FILE_MP3: u32 = 0;
    pub const RG_FILE_AAC: u32 = 1;
    pub const RG_TRACK_FLAG_NONFINITE: u32 = 1;
    pub const RG_TRACK_FLAG_IMPRECISE: u32 = 2;
    pub const RG_NODE_EXCHANGE_HOST: c_int = 0;
    pub const RG_NODE_EXCHANGE_RCCL: c_int = 1;

    /// `rg_ctx` (opaque)
    #[repr(C)]
    pub struct RgCtx {
        _p: [u8; 0],
    }
    /// `rg_node` (opaque)
    #[repr(C)]
    pub struct RgNode {
        _p: [u8; 0],
    }
    /// `rg_node_backend` (a test seam of the node header; only ever passed by pointer here)
    #[repr(C)]
    pub struct RgNodeBackend {
        _p: [u8; 0],
    }

    /// `rg_track_desc`: one decoded track inside a caller-owned planar PCM arena
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct RgTrackDesc {
        pub offset_bytes: u64,
        pub frames: u64,
        pub sample_rate: u32,
        pub channels: u16,
        pub format: u16,
    }

    /// `rg_track_result`: `ReplayGainResult` (src/replaygain.rs:57-68) + `gain_steps()` (:72-74)
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct RgTrackResult {
        pub loudness_db: f64,
        pub gain_db: f64,
        pub peak: f64,
        pub sample_rate: u32,
        pub gain_steps: i32,
        pub windows: u32,
        pub file_type: u32,
        pub flags: u32,
        pub reserved: u32,
    }

    /// `rg_album_result`: `AlbumGainResult` minus the per-track vector (src/replaygain.rs:79-95)
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct RgAlbumResult {
        pub album_loudness_db: f64,
        pub album_gain_db: f64,
        pub album_peak: f64,
        pub album_gain_steps: i32,
        pub windows: u32,
    }

    /// `rg_peak_result`: `PeakAmplitudeResult` (src/replaygain.rs:1125-1132)
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct RgPeakResult {
        pub peak: f64,
        pub peak_pcm: f64,
        pub sample_rate: u32,
        pub reserved: u32,
    }

    /// `rg_device_view`: device-side views for callers that keep results in HBM
    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct RgDeviceView {
        pub d_track_hist: *mut c_void,
        pub d_track_result: *mut c_void,
        pub d_album_hist: *mut c_void,
        pub d_album_peak: *mut c_void,
        pub n_tracks: u64,
    }

    /// `rg_wav_info`
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct RgWavInfo {
        pub sample_rate: u32,
        pub channels: u16,
        pub bits_per_sample: u16,
        pub sample_format: u16,
        pub block_align: u16,
        pub reserved: u32,
        pub data_offset: u64,
        pub frames: u64,
    }

    extern "C" {
        // ---- include/mp3rgain_amd.h
        pub fn rg_abi_version() -> c_int;
        pub fn rg_is_available() -> c_int;
        pub fn rg_supported_rate(sample_rate: u32) -> c_int;
        pub fn rg_window_samples(sample_rate: u32) -> u32;
        pub fn rg_hist_loudness(hist: *const u32) -> f64;
        pub fn rg_gain_from_loudness(loudness_db: f64) -> f64;
        pub fn rg_gain_steps(gain_db: f64) -> i32;
        pub fn rg_db_to_steps(db: f64) -> i32;
        pub fn rg_steps_to_db(steps: i32) -> f64;
        pub fn rg_clip_limit_steps(steps: i32, gain_db: f64, peak: f64, prevent_clipping: c_int, wrap_gain: c_int) -> i32;
        pub fn rg_rate_design_info(sample_rate: u32, stable: *mut c_int, halo_frames: *mut u32, decay_ratio: *mut f64) -> c_int;
        pub fn rg_create(device: c_int) -> *mut RgCtx;
        pub fn rg_destroy(ctx: *mut RgCtx);
        pub fn rg_last_error(ctx: *const RgCtx) -> *const c_char;
        pub fn rg_set_stream(ctx: *mut RgCtx, hip_stream: *mut c_void, attach: c_int) -> c_int;
        pub fn rg_wait_user_stream(ctx: *mut RgCtx) -> c_int;
        pub fn rg_batch_stream(ctx: *mut RgCtx) -> *mut c_void;
        pub fn rg_set_kernel(ctx: *mut RgCtx, variant: c_int) -> c_int;
        pub fn rg_set_tuning(ctx: *mut RgCtx, key: c_int, value: i64) -> c_int;
        pub fn rg_tm_design_info(sample_rate: u32, L: u32, H10: *mut u32, rounds: *mut u32, rounds_fast: *mut u32, decoupling_residual: *mut f64, T_out: *mut f64, gram_last_out: *mut f64) -> c_int;
        pub fn rg_tm_design_affine(sample_rate: u32, L: u32, servo: *mut c_int, alpha: *mut f64, beta: *mut f64, g: *mut f64, d_inf: *mut f64, sigma0_out: *mut f64) -> c_int;
        pub fn rg_analyze_pcm_batch(ctx: *mut RgCtx, tracks: *const RgTrackDesc, n: usize, pcm_base: *const c_void, pcm_bytes: usize, pcm_on_device: c_int, out: *mut RgTrackResult, hist_out: *mut u32) -> c_int;
        pub fn rg_analyze_album_pcm(ctx: *mut RgCtx, tracks: *const RgTrackDesc, n: usize, pcm_base: *const c_void, pcm_bytes: usize, pcm_on_device: c_int, tracks_out: *mut RgTrackResult, album_out: *mut RgAlbumResult, album_hist_out: *mut u32) -> c_int;
        pub fn rg_find_peak_pcm(ctx: *mut RgCtx, track: *const RgTrackDesc, pcm_base: *const c_void, pcm_bytes: usize, pcm_on_device: c_int, out: *mut RgPeakResult) -> c_int;
        pub fn rg_enqueue_pcm_batch(ctx: *mut RgCtx, tracks: *const RgTrackDesc, n: usize, d_pcm_base: *const c_void, pcm_bytes: usize, album: c_int) -> c_int;
        pub fn rg_device_view_get(ctx: *mut RgCtx, view: *mut RgDeviceView) -> c_int;
        pub fn rg_collect(ctx: *mut RgCtx, tracks_out: *mut RgTrackResult, hist_out: *mut u32) -> c_int;
        pub fn rg_collect_exact(ctx: *mut RgCtx, tracks: *const RgTrackDesc, n: usize, d_pcm_base: *const c_void, pcm_bytes: usize, tracks_out: *mut RgTrackResult, hist_out: *mut u32) -> c_int;
        pub fn rg_album_allreduce(ctx: *mut RgCtx, nccl_comm: *mut c_void) -> c_int;
        pub fn rg_comm_library(librccl_path: *const c_char) -> c_int;
        pub fn rg_comm_unique_id(id_out: *mut c_void) -> c_int;
        pub fn rg_comm_init(ctx: *mut RgCtx, id: *const c_void, world: c_int, rank: c_int) -> c_int;
        pub fn rg_comm_destroy(ctx: *mut RgCtx) -> c_int;
        pub fn rg_comm_info(ctx: *mut RgCtx, world_out: *mut c_int, version_out: *mut c_int) -> c_int;
        pub fn rg_album_exchange(ctx: *mut RgCtx) -> c_int;
        pub fn rg_album_reduce_gathered(ctx: *mut RgCtx, d_gathered: *const c_void, world: u32) -> c_int;
        pub fn rg_album_finish(ctx: *mut RgCtx, album_out: *mut RgAlbumResult, album_hist_out: *mut u32) -> c_int;
        pub fn rg_album_result_enqueue(ctx: *mut RgCtx) -> c_int;
        pub fn rg_timing_enable(ctx: *mut RgCtx, on: c_int) -> c_int;
        pub fn rg_timing_read(ctx: *mut RgCtx, sum_ms: *mut f64, launches: *mut u64, span_ms: *mut f64, reset: c_int) -> c_int;
        pub fn rg_synth_fill_device(ctx: *mut RgCtx, d_dst_f32: *mut c_void, seed: u64, channel: u32, sample_rate: u32, first_frame: u64, frames: u64) -> c_int;
        pub fn rg_wav_parse(data: *const c_void, len: usize, out: *mut RgWavInfo) -> c_int;
        pub fn rg_set_decoder_command(ctx: *mut RgCtx, command_template: *const c_char) -> c_int;
        pub fn rg_analyze_wav_batch(ctx: *mut RgCtx, wav: *const *const c_void, wav_len: *const usize, n: usize, album: c_int, out: *mut RgTrackResult, album_out: *mut RgAlbumResult) -> c_int;
        pub fn rg_analyze_track(ctx: *mut RgCtx, path: *const c_char, track_index: i32, out: *mut RgTrackResult) -> c_int;
        pub fn rg_analyze_tracks(ctx: *mut RgCtx, paths: *const *const c_char, n: usize, track_index: i32, out: *mut RgTrackResult, status_out: *mut i32) -> c_int;
        pub fn rg_tracks_error(ctx: *const RgCtx, i: usize) -> *const c_char;
        pub fn rg_analyze_album(ctx: *mut RgCtx, paths: *const *const c_char, n: usize, track_index: i32, tracks_out: *mut RgTrackResult, album_out: *mut RgAlbumResult) -> c_int;
        pub fn rg_analyze_album_begin(ctx: *mut RgCtx, paths: *const *const c_char, n: usize, track_index: i32, tracks_out: *mut RgTrackResult, failed_index: *mut usize) -> c_int;
        pub fn rg_find_peak_amplitude(ctx: *mut RgCtx, path: *const c_char, out: *mut RgPeakResult) -> c_int;
        pub fn rg_mp3_decode_device(ctx: *mut RgCtx, data: *const c_void, len: usize, ch0: *mut f32, ch1: *mut f32, capacity: u64, info: *mut c_void) -> c_int;
        pub fn rg_mp3_decode_bench(ctx: *mut RgCtx, data: *const c_void, len: usize, copies: u32, reps: u32, ms_out: *mut f64, units_out: *mut u64, compressed_bytes_out: *mut u64, frames_out: *mut u64) -> c_int;
        // ---- include/mp3rgain_amd_node.h
        pub fn rg_node_create(devices: *const c_int, n: usize) -> *mut RgNode;
        pub fn rg_node_destroy(node: *mut RgNode);
        pub fn rg_node_last_error(node: *const RgNode) -> *const c_char;
        pub fn rg_node_devices(node: *const RgNode) -> usize;
        pub fn rg_node_ctx(node: *mut RgNode, i: usize) -> *mut RgCtx;
        pub fn rg_node_set_exchange(node: *mut RgNode, mode: c_int) -> c_int;
        pub fn rg_node_partition(sizes: *const u64, n: usize, world: usize, owner_out: *mut u32);
        pub fn rg_analyze_album_node(node: *mut RgNode, paths: *const *const c_char, n: usize, track_index: i32, tracks_out: *mut RgTrackResult, album_out: *mut RgAlbumResult) -> c_int;
        pub fn rg_analyze_tracks_node(node: *mut RgNode, paths: *const *const c_char, n: usize, track_index: i32, out: *mut RgTrackResult, status_out: *mut i32) -> c_int;
        pub fn rg_node_tracks_error(node: *const RgNode, i: usize) -> *const c_char;
        pub fn rg_node_last_partition(node: *const RgNode, owner_out: *mut u32, n: usize) -> c_int;
        pub fn rg_node_create_backend(backend: *const RgNodeBackend, devices: *const c_int, n: usize) -> *mut RgNode;
    }
}

pub mod replaygain {
    //! mp3rgain's `replaygain` module (src/replaygain.rs), same names and signatures, run on the GPUs of the node.
    use super::ffi;
    use anyhow::{anyhow, bail, Result};
    use std::ffi::{CStr, CString};
    use std::os::raw::{c_char, c_void};
    use std::os::unix::ffi::OsStrExt;
    use std::path::Path;
    use std::sync::{Mutex, OnceLock};

    /// GAIN_STEP_DB, src/lib.rs:48
    pub const GAIN_STEP_DB: f64 = 1.5;

    /// src/replaygain.rs:48-53
    #[derive(Debug, Clone, Copy, PartialEq)]
    pub enum AudioFileType {
        Mp3,
        Aac,
    }

    /// src/replaygain.rs:57-68
    #[derive(Debug, Clone)]
    pub struct ReplayGainResult {
        pub loudness_db: f64,
        pub gain_db: f64,
        pub peak: f64,
        pub sample_rate: u32,
        pub file_type: AudioFileType,
    }

    impl ReplayGainResult {
        /// src/replaygain.rs:72-74
        pub fn gain_steps(&self) -> i32 {
            (self.gain_db / GAIN_STEP_DB).round() as i32
        }
    }

    /// src/replaygain.rs:79-88
    #[derive(Debug, Clone)]
    pub struct AlbumGainResult {
        pub tracks: Vec<ReplayGainResult>,
        pub album_loudness_db: f64,
        pub album_gain_db: f64,
        pub album_peak: f64,
    }

    impl AlbumGainResult {
        /// src/replaygain.rs:92-94
        pub fn album_gain_steps(&self) -> i32 {
            (self.album_gain_db / GAIN_STEP_DB).round() as i32
        }
    }

    /// src/replaygain.rs:1125-1132
    #[derive(Debug, Clone)]
    pub struct PeakAmplitudeResult {
        pub peak: f64,
        pub peak_pcm: f64,
        pub sample_rate: u32,
    }

    fn to_result(r: &ffi::RgTrackResult) -> ReplayGainResult {
        ReplayGainResult {
            loudness_db: r.loudness_db,
            gain_db: r.gain_db,
            peak: r.peak,
            sample_rate: r.sample_rate,
            file_type: if r.file_type == ffi::RG_FILE_AAC { AudioFileType::Aac } else { AudioFileType::Mp3 },
        }
    }

    unsafe fn text(p: *const c_char) -> String {
        if p.is_null() {
            String::from("mp3rgain_amd: unknown error")
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }

    /// All GPUs of the node behind one handle (`rg_node`, include/mp3rgain_amd_node.h): one context and one host thread
    /// per device inside the library.  Calls are serialised by the mutex of [`node`].
    pub struct Node {
        raw: *mut ffi::RgNode,
    }
    // The library's node serialises nothing itself; this binding only ever touches it under the mutex below.
    unsafe impl Send for Node {}

    impl Node {
        /// `devices`: HIP ordinals; empty = every visible device.
        pub fn new(devices: &[i32]) -> Result<Node> {
            if unsafe { ffi::rg_abi_version() } != ffi::RG_ABI_VERSION {
                bail!("libmp3rgain_amd.so has ABI {}, this binding was written for {}", unsafe { ffi::rg_abi_version() }, ffi::RG_ABI_VERSION);
            }
            let raw = unsafe {
                if devices.is_empty() { ffi::rg_node_create(std::ptr::null(), 0) } else { ffi::rg_node_create(devices.as_ptr(), devices.len()) }
            };
            if raw.is_null() {
                bail!("{}", unsafe { text(ffi::rg_node_last_error(std::ptr::null())) });
            }
            Ok(Node { raw })
        }
        fn error(&self) -> anyhow::Error {
            anyhow!("{}", unsafe { text(ffi::rg_node_last_error(self.raw)) })
        }
        /// The context of device `i`, for `rg_set_tuning` / `rg_set_kernel` / the PCM-level calls.
        pub fn ctx(&self, i: usize) -> *mut ffi::RgCtx {
            unsafe { ffi::rg_node_ctx(self.raw, i) }
        }
        pub fn devices(&self) -> usize {
            unsafe { ffi::rg_node_devices(self.raw) }
        }
    }

    impl Drop for Node {
        fn drop(&mut self) {
            unsafe { ffi::rg_node_destroy(self.raw) }
        }
    }

    static NODE: OnceLock<Mutex<Node>> = OnceLock::new();

    /// The process-wide node, created on first use (`MP3RGAIN_AMD_DEVICES=0,2` restricts it, as for the Python CLI).
    pub fn node() -> Result<&'static Mutex<Node>> {
        if let Some(n) = NODE.get() {
            return Ok(n);
        }
        let devices: Vec<i32> = std::env::var("MP3RGAIN_AMD_DEVICES")
            .ok()
            .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect())
            .unwrap_or_default();
        let n = Node::new(&devices)?;
        Ok(NODE.get_or_init(|| Mutex::new(n)))
    }

    fn c_paths(files: &[&Path]) -> Result<(Vec<CString>, Vec<*const c_char>)> {
        let owned: Vec<CString> = files
            .iter()
            .map(|p| CString::new(p.as_os_str().as_bytes()).map_err(|_| anyhow!("Failed to open: {}", p.display())))
            .collect::<Result<_>>()?;
        let ptrs = owned.iter().map(|s| s.as_ptr()).collect();
        Ok((owned, ptrs))
    }

    fn index_arg(track_index: Option<u32>) -> i32 {
        track_index.map_or(-1, |i| i as i32)
    }

    /// src/replaygain.rs:1119-1121 -- true when the library answers and a gfx950 device is usable (there is no CPU path)
    pub fn is_available() -> bool {
        unsafe { ffi::rg_is_available() != 0 }
    }

    /// Decoder command for files the library does not decode itself (M4A/AAC): a shell template writing WAV to stdout,
    /// `{}` = the quoted path; applied to every device's context.
    pub fn set_decoder_command(command_template: &str) -> Result<()> {
        let c = CString::new(command_template)?;
        let n = node()?.lock().unwrap();
        for i in 0..n.devices() {
            if unsafe { ffi::rg_set_decoder_command(n.ctx(i), c.as_ptr()) } != ffi::RG_OK {
                bail!("{}", unsafe { text(ffi::rg_last_error(n.ctx(i))) });
            }
        }
        Ok(())
    }

    /// src/replaygain.rs:929-932
    pub fn analyze_track(file_path: &Path) -> Result<ReplayGainResult> {
        analyze_track_with_index(file_path, None)
    }

    /// src/replaygain.rs:935-941
    pub fn analyze_track_with_index(file_path: &Path, track_index: Option<u32>) -> Result<ReplayGainResult> {
        let mut v = analyze_tracks_with_index(&[file_path], track_index)?;
        v.remove(0)
    }

    /// The `-r` loop of src/main.rs:1937-2001 as one call: every file's own `Result`, all files decoded and analysed as
    /// one batch per GPU (replicas only; no exchange).
    pub fn analyze_tracks_with_index(files: &[&Path], track_index: Option<u32>) -> Result<Vec<Result<ReplayGainResult>>> {
        let (_owned, ptrs) = c_paths(files)?;
        let mut out = vec![ffi::RgTrackResult::default(); files.len()];
        let mut status = vec![0i32; files.len()];
        let n = node()?.lock().unwrap();
        let rc = unsafe { ffi::rg_analyze_tracks_node(n.raw, ptrs.as_ptr(), ptrs.len(), index_arg(track_index), out.as_mut_ptr(), status.as_mut_ptr()) };
        if rc != ffi::RG_OK {
            return Err(n.error());
        }
        Ok((0..files.len())
            .map(|i| {
                if status[i] == ffi::RG_OK {
                    Ok(to_result(&out[i]))
                } else {
                    Err(anyhow!("{}", unsafe { text(ffi::rg_node_tracks_error(n.raw, i)) }))
                }
            })
            .collect())
    }

    /// src/replaygain.rs:1033-1036
    pub fn analyze_album(files: &[&Path]) -> Result<AlbumGainResult> {
        analyze_album_with_index(files, None)
    }

    /// src/replaygain.rs:1044-1074: the files are dealt to the node's GPUs by size, the first failing file in input order
    /// is the album's error (`:1055`), the devices' histograms and peaks are folded, results come back in input order.
    pub fn analyze_album_with_index(files: &[&Path], track_index: Option<u32>) -> Result<AlbumGainResult> {
        let (_owned, ptrs) = c_paths(files)?;
        let mut tracks = vec![ffi::RgTrackResult::default(); files.len()];
        let mut album = ffi::RgAlbumResult::default();
        let n = node()?.lock().unwrap();
        let rc = unsafe { ffi::rg_analyze_album_node(n.raw, ptrs.as_ptr(), ptrs.len(), index_arg(track_index), tracks.as_mut_ptr(), &mut album) };
        if rc != ffi::RG_OK {
            return Err(n.error());
        }
        Ok(AlbumGainResult {
            tracks: tracks.iter().map(to_result).collect(),
            album_loudness_db: album.album_loudness_db,
            album_gain_db: album.album_gain_db,
            album_peak: album.album_peak,
        })
    }

    /// src/replaygain.rs:1140-1249
    pub fn find_peak_amplitude(file_path: &Path) -> Result<PeakAmplitudeResult> {
        let (_owned, ptrs) = c_paths(&[file_path])?;
        let mut out = ffi::RgPeakResult::default();
        let n = node()?.lock().unwrap();
        let ctx = n.ctx(0);
        if unsafe { ffi::rg_find_peak_amplitude(ctx, ptrs[0], &mut out) } != ffi::RG_OK {
            bail!("{}", unsafe { text(ffi::rg_last_error(ctx)) });
        }
        Ok(PeakAmplitudeResult { peak: out.peak, peak_pcm: out.peak_pcm, sample_rate: out.sample_rate })
    }

    /// One decoded track for the seam after the decoder: what `process_audio_buffer` (src/replaygain.rs:953-1029) sees of
    /// a file -- channel 0, channel 1 if there is one (only these two are read, `:971`), the rate.  f32, normalised.
    pub struct DecodedTrack {
        pub left: Vec<f32>,
        pub right: Option<Vec<f32>>,
        pub sample_rate: u32,
        pub file_type: AudioFileType,
    }

    fn arena_of(decoded: &[DecodedTrack]) -> (Vec<f32>, Vec<ffi::RgTrackDesc>) {
        let mut arena: Vec<f32> = Vec::new();
        let mut descs = Vec::with_capacity(decoded.len());
        for t in decoded {
            descs.push(ffi::RgTrackDesc {
                offset_bytes: (arena.len() * 4) as u64,
                frames: t.left.len() as u64,
                sample_rate: t.sample_rate,
                channels: if t.right.is_some() { 2 } else { 1 },
                format: ffi::RG_FMT_F32_PLANAR,
            });
            arena.extend_from_slice(&t.left);
            if let Some(r) = &t.right {
                arena.extend_from_slice(r);
            }
        }
        (arena, descs)
    }

    /// `analyze_track_internal` from the filters onwards (src/replaygain.rs:866-925) for n decoded tracks, on device 0.
    pub fn analyze_tracks_pcm(decoded: &[DecodedTrack]) -> Result<Vec<ReplayGainResult>> {
        let (arena, descs) = arena_of(decoded);
        let mut out = vec![ffi::RgTrackResult::default(); descs.len()];
        let n = node()?.lock().unwrap();
        let ctx = n.ctx(0);
        let rc = unsafe {
            ffi::rg_analyze_pcm_batch(ctx, descs.as_ptr(), descs.len(), arena.as_ptr() as *const c_void, arena.len() * 4, 0, out.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != ffi::RG_OK {
            bail!("{}", unsafe { text(ffi::rg_last_error(ctx)) });
        }
        Ok(out.iter().zip(decoded).map(|(r, d)| ReplayGainResult { file_type: d.file_type, ..to_result(r) }).collect())
    }

    /// The loop of `analyze_album_with_index` (src/replaygain.rs:1053-1074) over decoded tracks, on device 0.
    pub fn analyze_album_pcm(decoded: &[DecodedTrack]) -> Result<AlbumGainResult> {
        let (arena, descs) = arena_of(decoded);
        let mut tracks = vec![ffi::RgTrackResult::default(); descs.len()];
        let mut album = ffi::RgAlbumResult::default();
        let n = node()?.lock().unwrap();
        let ctx = n.ctx(0);
        let rc = unsafe {
            ffi::rg_analyze_album_pcm(ctx, descs.as_ptr(), descs.len(), arena.as_ptr() as *const c_void, arena.len() * 4, 0, tracks.as_mut_ptr(), &mut album, std::ptr::null_mut())
        };
        if rc != ffi::RG_OK {
            bail!("{}", unsafe { text(ffi::rg_last_error(ctx)) });
        }
        Ok(AlbumGainResult {
            tracks: tracks.iter().zip(decoded).map(|(r, d)| ReplayGainResult { file_type: d.file_type, ..to_result(r) }).collect(),
            album_loudness_db: album.album_loudness_db,
            album_gain_db: album.album_gain_db,
            album_peak: album.album_peak,
        })
    }
}
