// TEST INFRASTRUCTURE, not product code: a stand-in for the handful of librccl.so entry points the library resolves at run
// time (mp3rgain_amd/csrc/rg_capi.hip: resolve), handed to it through rg_comm_library().
//
// Why it exists: the GPU box of this build is ONE MI355X, and RCCL refuses a communicator with two ranks on one device
// ("Duplicate GPU detected"), so the library's multi-rank code -- rg_comm_init / rg_comm_init_all with world > 1, the
// all-gather of the [histogram | peak] packs on the batch's stream, the fold of SEVERAL different packs, bench.py's
// world > 1 branch -- could never execute before the driver's 8-GPU run.  With this transport in RCCL's place it does:
// several ranks (host threads of one process, or separate processes) share the leased device, everything above the
// collective is the real code.  The transport itself proves nothing about RCCL or xGMI and is never timed.
//
// Transport: a POSIX shared-memory block named by the unique id.  A collective is D2H of the rank's part into the block
// on the caller's stream, a barrier, H2D of the whole block, a second barrier (nobody overwrites the block while another
// rank still reads it).  Every wait has a deadline: a rank that never arrives makes the others return an error instead of
// hanging the box.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr size_t kIdBytes = 128;          // ncclUniqueId
constexpr size_t kDataBytes = 4u << 20;   // staging for one collective (8 ranks x 48 KB packs need 0.4 MB)
constexpr int kMaxRanks = 64;
constexpr double kDeadlineSeconds = 60.0;

struct Shared {
    std::atomic<uint32_t> arrived;     // barrier: ranks that reached the current generation
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> joined;      // ranks that mapped the block (init) / left it (destroy)
    uint32_t pad;
    alignas(64) unsigned char data[kDataBytes];
};

struct Comm {
    Shared *sh;
    int world, rank;
    char name[kIdBytes];
    bool in_process;  // ncclCommInitAll: the block is plain heap memory shared by the threads of this process
    std::atomic<int> *in_process_refs;
};

enum { kSuccess = 0, kSystemError = 2, kInvalidArgument = 4, kInvalidUsage = 5 };

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool barrier(Comm *c) {
    Shared *s = c->sh;
    const uint32_t gen = s->generation.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const double t0 = now();
    while (s->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (now() - t0 > kDeadlineSeconds) return false;
    }
    return true;
}

size_t type_bytes(int dtype) {
    switch (dtype) {  // rccl.h: ncclDataType_t
        case 0: case 1: return 1;           // int8 / uint8
        case 2: case 3: case 7: return 4;   // int32 / uint32 / float32
        case 4: case 5: case 8: return 8;   // int64 / uint64 / float64
        case 6: case 9: return 2;           // float16 / bfloat16
        default: return 0;
    }
}

std::atomic<uint32_t> g_counter{0};

}  // namespace

extern "C" {

const char *ncclGetErrorString(int r) {
    switch (r) {
        case kSuccess: return "no error (stand-in transport)";
        case kSystemError: return "stand-in transport: a rank did not arrive within the deadline, or shared memory failed";
        case kInvalidArgument: return "stand-in transport: invalid argument";
        case kInvalidUsage: return "stand-in transport: invalid usage";
        default: return "stand-in transport: error";
    }
}

int ncclGetUniqueId(void *id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, kIdBytes);
    snprintf(static_cast<char *>(id), kIdBytes, "/rg_standin_%ld_%u_%llx", (long)getpid(), g_counter.fetch_add(1),
             (unsigned long long)(now() * 1e6));
    return kSuccess;
}

struct IdByValue { char bytes[kIdBytes]; };

int ncclCommInitRank(void **comm_out, int world, IdByValue id, int rank) {
    if (!comm_out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return kInvalidArgument;
    id.bytes[kIdBytes - 1] = 0;
    if (id.bytes[0] != '/') return kInvalidArgument;
    const int fd = shm_open(id.bytes, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return kSystemError;
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return kSystemError; }  // a fresh object is zero-filled
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return kSystemError;
    Comm *c = new Comm{static_cast<Shared *>(p), world, rank, {0}, false, nullptr};
    memcpy(c->name, id.bytes, kIdBytes);
    c->sh->joined.fetch_add(1, std::memory_order_acq_rel);
    // like ncclCommInitRank: returns once every rank has joined
    const double t0 = now();
    while (c->sh->joined.load(std::memory_order_acquire) < (uint32_t)world) {
        sched_yield();
        if (now() - t0 > kDeadlineSeconds) {
            munmap(p, sizeof(Shared));
            shm_unlink(id.bytes);
            delete c;
            return kSystemError;
        }
    }
    *comm_out = c;
    return kSuccess;
}

// ranks = positions in devs; unlike RCCL, the same device may appear more than once (that is the point)
int ncclCommInitAll(void **comms, int n, const int *devs) {
    (void)devs;
    if (!comms || n < 1 || n > kMaxRanks) return kInvalidArgument;
    Shared *s = static_cast<Shared *>(aligned_alloc(64, sizeof(Shared)));
    if (!s) return kSystemError;
    memset(static_cast<void *>(s), 0, sizeof(Shared));
    std::atomic<int> *refs = new std::atomic<int>(n);
    for (int i = 0; i < n; ++i) comms[i] = new Comm{s, n, i, {0}, true, refs};
    return kSuccess;
}

int ncclCommDestroy(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return kInvalidArgument;
    if (c->in_process) {
        if (c->in_process_refs->fetch_sub(1) == 1) {
            free(c->sh);
            delete c->in_process_refs;
        }
    } else {
        munmap(c->sh, sizeof(Shared));
        shm_unlink(c->name);  // the first rank to leave removes the name; the mappings of the others stay valid
    }
    delete c;
    return kSuccess;
}

int ncclGroupStart(void) { return kSuccess; }
int ncclGroupEnd(void) { return kSuccess; }

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t bytes = count * type_bytes(dtype);
    if (!c || !send || !recv || bytes == 0 || bytes * (size_t)c->world > kDataBytes) return kInvalidArgument;
    if (hipMemcpyAsync(c->sh->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return kSystemError;
    if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;
    if (!barrier(c)) return kSystemError;
    if (hipMemcpyAsync(recv, c->sh->data, bytes * (size_t)c->world, hipMemcpyHostToDevice, stream) != hipSuccess) return kSystemError;
    if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;
    if (!barrier(c)) return kSystemError;
    return kSuccess;
}

// sum of uint32 / int32 (wrapping) and max of float64: what rg_album_allreduce asks for
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    const size_t bytes = count * type_bytes(dtype);
    if (!c || !send || !recv || bytes == 0 || bytes * (size_t)c->world > kDataBytes) return kInvalidArgument;
    const bool sum_u32 = (dtype == 2 || dtype == 3) && op == 0, max_f64 = dtype == 8 && op == 2;
    if (!sum_u32 && !max_f64) return kInvalidUsage;
    if (hipMemcpyAsync(c->sh->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return kSystemError;
    if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;
    if (!barrier(c)) return kSystemError;
    unsigned char *out = static_cast<unsigned char *>(malloc(bytes));
    if (!out) return kSystemError;
    memcpy(out, c->sh->data, bytes);
    for (int r = 1; r < c->world; ++r) {
        const unsigned char *src = c->sh->data + (size_t)r * bytes;
        for (size_t i = 0; i < count; ++i) {
            if (sum_u32) {
                uint32_t a, b;
                memcpy(&a, out + 4 * i, 4); memcpy(&b, src + 4 * i, 4);
                a += b;
                memcpy(out + 4 * i, &a, 4);
            } else {
                double a, b;
                memcpy(&a, out + 8 * i, 8); memcpy(&b, src + 8 * i, 8);
                if (b > a) a = b;
                memcpy(out + 8 * i, &a, 8);
            }
        }
    }
    const bool ok = hipMemcpyAsync(recv, out, bytes, hipMemcpyHostToDevice, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    free(out);
    if (!barrier(c)) return kSystemError;
    return ok ? kSuccess : kSystemError;
}

}  // extern "C"
