// rg_node.hip -- include/mp3rgain_amd_node.h: the file-level entry points over all GPUs of one node, one process.
//
// analyze_album_with_index (src/replaygain.rs:1044-1074) is one blocking call; here it stays one blocking call and the
// node's GPUs share the work: files are dealt out by size, every device's share runs through its own context on a host
// thread of its own (file read, device MP3 decode, the analysis kernels), the devices' [histogram | peak] packs are merged
// (LoudnessHistogram::accumulate :658-662, album_peak.max :1056) and the percentile is read off the merged histogram.
// Host code only; everything device-side happens behind the per-device engine (rg_ctx by default).
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mp3rgain_amd_node.h"
#include "rg_ctx.h"

namespace {
thread_local std::string g_node_create_error;  // what rg_node_last_error(NULL) returns to the thread whose create failed

// ---- the built-in engine: one rg_ctx -------------------------------------------------------------------------------
void *ctx_open(int device, void *) { return rg_create(device); }
void ctx_close(void *e, void *) { rg_destroy(static_cast<rg_ctx *>(e)); }
int ctx_album_begin(void *e, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out, size_t *failed, void *) {
    return rg_analyze_album_begin(static_cast<rg_ctx *>(e), paths, n, track_index, out, failed);
}
int ctx_album_pack(void *e, uint32_t *pack, void *) {
    rg_album_result alb;
    const int rc = rg_album_finish(static_cast<rg_ctx *>(e), &alb, pack);
    if (rc != RG_OK) return rc;
    memcpy(pack + RG_HISTOGRAM_SIZE, &alb.album_peak, sizeof(double));
    return RG_OK;
}
int ctx_tracks(void *e, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out, int32_t *status, void *) {
    return rg_analyze_tracks(static_cast<rg_ctx *>(e), paths, n, track_index, out, status);
}
const char *ctx_tracks_error(void *e, size_t i, void *) { return rg_tracks_error(static_cast<rg_ctx *>(e), i); }
const char *ctx_last_error(void *e, void *) { return rg_last_error(static_cast<rg_ctx *>(e)); }
const rg_node_backend kCtxBackend = {ctx_open, ctx_close, ctx_album_begin, ctx_album_pack, ctx_tracks, ctx_tracks_error, ctx_last_error, nullptr};
}  // namespace

struct rg_node {
    rg_node_backend be{};
    bool builtin = false;
    std::vector<int> devices;
    std::vector<void *> engines;
    int exchange = RG_NODE_EXCHANGE_HOST;
    std::string err;
    std::vector<std::string> track_errors;
    std::vector<uint32_t> owner;  // the last call's dealing
};

namespace {
int node_err(rg_node *nd, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    nd->err = buf;
    return code;
}

// files by size (a proxy for their length), dealt out; shares[d] = indices of device d's files, ascending
void deal(rg_node *nd, const char *const *paths, size_t n, std::vector<std::vector<size_t>> *shares) {
    std::vector<uint64_t> sizes(n, 0);
    for (size_t i = 0; i < n; ++i) {
        struct stat st;
        if (paths[i] && stat(paths[i], &st) == 0 && st.st_size > 0) sizes[i] = (uint64_t)st.st_size;  // a missing file: its owner reports it
    }
    nd->owner.assign(n, 0);
    rg_node_partition(sizes.data(), n, nd->engines.size(), nd->owner.data());
    shares->assign(nd->engines.size(), std::vector<size_t>());
    for (size_t i = 0; i < n; ++i) (*shares)[nd->owner[i]].push_back(i);
}

template <typename F>
void on_every_device(size_t n_dev, F f) {
    if (n_dev == 1) {
        f(0);
        return;
    }
    std::vector<std::thread> pool;
    for (size_t d = 0; d < n_dev; ++d) pool.emplace_back(f, d);
    for (auto &t : pool) t.join();
}
}  // namespace

extern "C" void rg_node_partition(const uint64_t *sizes, size_t n, size_t world, uint32_t *owner_out) {
    if (!owner_out || world == 0) return;
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return (sizes ? sizes[a] : 0) > (sizes ? sizes[b] : 0); });
    std::vector<uint64_t> load(world, 0);
    for (size_t i : order) {
        size_t best = 0;
        for (size_t d = 1; d < world; ++d)
            if (load[d] < load[best]) best = d;
        owner_out[i] = (uint32_t)best;
        load[best] += sizes ? sizes[i] : 0;
    }
}

extern "C" rg_node *rg_node_create_backend(const rg_node_backend *backend, const int *devices, size_t n) {
    if (!backend || !backend->open || !backend->close || !backend->album_begin || !backend->album_pack || !backend->tracks ||
        !backend->tracks_error || !backend->last_error || !devices || n == 0) {
        g_node_create_error = "rg_node_create_backend: incomplete backend or empty device list";
        return nullptr;
    }
    if (backend != &kCtxBackend) {  // engines other than the library's own contexts: a test seam (tests/test_node_cpu.py)
        const char *seams = getenv("MP3RGAIN_AMD_TEST_SEAMS");
        if (!(seams && seams[0] == '1')) {
            g_node_create_error = "rg_node_create_backend: foreign engines are a test seam (MP3RGAIN_AMD_TEST_SEAMS=1)";
            return nullptr;
        }
    }
    rg_node *nd = new rg_node;
    nd->be = *backend;
    nd->builtin = backend == &kCtxBackend;
    nd->devices.assign(devices, devices + n);
    for (size_t i = 0; i < n; ++i) {
        void *e = nd->be.open(devices[i], nd->be.user);
        if (!e) {
            g_node_create_error = nd->builtin ? std::string(rg_last_error(nullptr)) : std::string("device ") + std::to_string(devices[i]) + ": the backend could not open it";
            rg_node_destroy(nd);
            return nullptr;
        }
        nd->engines.push_back(e);
    }
    if (nd->builtin && n > 1) {  // the file loaders of all contexts share the host's cores
        const unsigned share = std::max(1u, rg_usable_cores() / (unsigned)n);
        for (void *e : nd->engines) (void)rg_set_tuning(static_cast<rg_ctx *>(e), RG_TUNE_LOADER_THREADS, share);
    }
    return nd;
}

extern "C" rg_node *rg_node_create(const int *devices, size_t n) {
    std::vector<int> all;
    if (!devices) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            g_node_create_error = "no usable gfx950 device (there is no CPU path)";
            return nullptr;
        }
        for (int d = 0; d < count; ++d) all.push_back(d);
        devices = all.data();
        n = all.size();
    }
    if (n == 0) {
        g_node_create_error = "rg_node_create: empty device list";
        return nullptr;
    }
    return rg_node_create_backend(&kCtxBackend, devices, n);
}

extern "C" void rg_node_destroy(rg_node *nd) {
    if (!nd) return;
    for (void *e : nd->engines) nd->be.close(e, nd->be.user);
    delete nd;
}

extern "C" const char *rg_node_last_error(const rg_node *nd) { return nd ? nd->err.c_str() : g_node_create_error.c_str(); }
extern "C" size_t rg_node_devices(const rg_node *nd) { return nd ? nd->engines.size() : 0; }
extern "C" rg_ctx *rg_node_ctx(rg_node *nd, size_t i) {
    return (nd && nd->builtin && i < nd->engines.size()) ? static_cast<rg_ctx *>(nd->engines[i]) : nullptr;
}

extern "C" int rg_node_set_exchange(rg_node *nd, int mode) {
    if (!nd) return RG_ERR_INVALID_ARG;
    if (mode == RG_NODE_EXCHANGE_HOST) {
        if (nd->builtin)
            for (void *e : nd->engines) (void)rg_comm_destroy(static_cast<rg_ctx *>(e));
        nd->exchange = mode;
        return RG_OK;
    }
    if (mode != RG_NODE_EXCHANGE_RCCL) return node_err(nd, RG_ERR_INVALID_ARG, "unknown exchange mode %d", mode);
    if (!nd->builtin) return node_err(nd, RG_ERR_INVALID_ARG, "the RCCL exchange needs the built-in engine (rg_ctx)");
    std::vector<rg_ctx *> ctxs;
    for (void *e : nd->engines) ctxs.push_back(static_cast<rg_ctx *>(e));
    const int rc = rg_comm_init_all(ctxs.data(), ctxs.size());
    if (rc != RG_OK) return node_err(nd, rc, "%s", rg_last_error(ctxs[0]));
    nd->exchange = mode;
    return RG_OK;
}

extern "C" int rg_node_last_partition(const rg_node *nd, uint32_t *owner_out, size_t n) {
    if (!nd || !owner_out || n != nd->owner.size()) return RG_ERR_INVALID_ARG;
    memcpy(owner_out, nd->owner.data(), n * sizeof(uint32_t));
    return RG_OK;
}

extern "C" int rg_analyze_album_node(rg_node *nd, const char *const *paths, size_t n, int32_t track_index, rg_track_result *tracks_out,
                                     rg_album_result *album_out) {
    if (!nd || (n && (!paths || !tracks_out)) || !album_out) return RG_ERR_INVALID_ARG;
    const size_t D = nd->engines.size();
    std::vector<std::vector<size_t>> shares;
    deal(nd, paths, n, &shares);
    struct Share {
        std::vector<const char *> paths;
        std::vector<rg_track_result> out;
        int rc = RG_OK;
        size_t failed = (size_t)-1;  // index into the call's paths
        std::string err;
        std::vector<uint32_t> pack;
        rg_album_result alb{};
    };
    std::vector<Share> sh(D);
    for (size_t d = 0; d < D; ++d) {
        for (size_t i : shares[d]) sh[d].paths.push_back(paths[i]);
        sh[d].out.resize(shares[d].size());
    }
    // ---- every device: its share of the album up to the percentile -----------------------------------------------
    on_every_device(D, [&](size_t d) {
        Share &s = sh[d];
        size_t failed = (size_t)-1;
        s.rc = nd->be.album_begin(nd->engines[d], s.paths.data(), s.paths.size(), track_index, s.out.data(), &failed, nd->be.user);
        if (s.rc != RG_OK) {
            s.err = nd->be.last_error(nd->engines[d], nd->be.user);
            if (failed < shares[d].size()) s.failed = shares[d][failed];
        }
    });
    // ---- nobody enters the exchange unless everybody got here: the first failing file in input order is the album's
    // error (`?` at src/replaygain.rs:1055); a failure that is not a file's (a device error) is reported if no file failed
    {
        const Share *worst = nullptr;
        for (const Share &s : sh)
            if (s.rc != RG_OK && (!worst || s.failed < worst->failed)) worst = &s;
        if (worst) return node_err(nd, worst->rc, "%s", worst->err.c_str());
    }
    // ---- the merge -------------------------------------------------------------------------------------------
    if (nd->exchange == RG_NODE_EXCHANGE_RCCL) {
        on_every_device(D, [&](size_t d) {
            rg_ctx *c = static_cast<rg_ctx *>(nd->engines[d]);
            sh[d].rc = rg_album_exchange(c);
            if (sh[d].rc == RG_OK) sh[d].rc = rg_album_finish(c, &sh[d].alb, nullptr);
            if (sh[d].rc != RG_OK) sh[d].err = rg_last_error(c);
        });
        for (size_t d = 0; d < D; ++d)
            if (sh[d].rc != RG_OK) return node_err(nd, sh[d].rc, "device %d: %s", nd->devices[d], sh[d].err.c_str());
        *album_out = sh[0].alb;
    } else {
        on_every_device(D, [&](size_t d) {
            sh[d].pack.assign(RG_ALBUM_PACK_WORDS, 0u);
            sh[d].rc = nd->be.album_pack(nd->engines[d], sh[d].pack.data(), nd->be.user);
            if (sh[d].rc != RG_OK) sh[d].err = nd->be.last_error(nd->engines[d], nd->be.user);
        });
        for (size_t d = 0; d < D; ++d)
            if (sh[d].rc != RG_OK) return node_err(nd, sh[d].rc, "device %d: %s", nd->devices[d], sh[d].err.c_str());
        std::vector<uint32_t> hist(RG_HISTOGRAM_SIZE, 0u);
        double peak = 0.0;
        uint64_t total = 0;
        for (size_t d = 0; d < D; ++d) {
            for (size_t b = 0; b < RG_HISTOGRAM_SIZE; ++b) hist[b] += sh[d].pack[b];  // u32, wrapping: the reference's release build
            double p;
            memcpy(&p, sh[d].pack.data() + RG_HISTOGRAM_SIZE, sizeof p);
            if (p > peak) peak = p;
        }
        for (uint32_t v : hist) total += v;
        rg_album_result r{};
        r.album_loudness_db = rg_hist_loudness(hist.data());
        r.album_gain_db = rg_gain_from_loudness(r.album_loudness_db);
        r.album_peak = peak;
        r.album_gain_steps = rg_gain_steps(r.album_gain_db);
        r.windows = (uint32_t)total;
        *album_out = r;
    }
    for (size_t d = 0; d < D; ++d)
        for (size_t k = 0; k < shares[d].size(); ++k) tracks_out[shares[d][k]] = sh[d].out[k];
    return RG_OK;
}

extern "C" int rg_analyze_tracks_node(rg_node *nd, const char *const *paths, size_t n, int32_t track_index, rg_track_result *out,
                                      int32_t *status_out) {
    if (!nd || (n && (!paths || !out || !status_out))) return RG_ERR_INVALID_ARG;
    const size_t D = nd->engines.size();
    std::vector<std::vector<size_t>> shares;
    deal(nd, paths, n, &shares);
    nd->track_errors.assign(n, std::string());
    std::vector<int> rcs(D, RG_OK);
    on_every_device(D, [&](size_t d) {
        const std::vector<size_t> &mine = shares[d];
        if (mine.empty()) return;
        std::vector<const char *> p;
        for (size_t i : mine) p.push_back(paths[i]);
        std::vector<rg_track_result> res(mine.size());
        std::vector<int32_t> st(mine.size(), RG_OK);
        rcs[d] = nd->be.tracks(nd->engines[d], p.data(), p.size(), track_index, res.data(), st.data(), nd->be.user);
        for (size_t k = 0; k < mine.size(); ++k) {
            if (rcs[d] != RG_OK) {  // the call itself failed (a device error): every file of the share carries it
                memset(&out[mine[k]], 0, sizeof out[0]);
                status_out[mine[k]] = rcs[d];
                nd->track_errors[mine[k]] = nd->be.last_error(nd->engines[d], nd->be.user);
                continue;
            }
            out[mine[k]] = res[k];
            status_out[mine[k]] = st[k];
            if (st[k] != RG_OK) nd->track_errors[mine[k]] = nd->be.tracks_error(nd->engines[d], k, nd->be.user);
        }
    });
    return RG_OK;
}

extern "C" const char *rg_node_tracks_error(const rg_node *nd, size_t i) {
    if (!nd || i >= nd->track_errors.size()) return "";
    return nd->track_errors[i].c_str();
}
