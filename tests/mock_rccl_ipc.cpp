/* Process-per-rank test double of the RCCL entry points libbfhip binds with dlsym (stwo-brainfuck_amd/csrc/comm.hip: RcclApi), selected with
 * the environment variable BFHIP_RCCL_LIBRARY. Unlike tests/mock_rccl.c (ranks = threads, host buffers) the ranks here are PROCESSES, each
 * with its own HIP context, and the buffers are DEVICE memory: what a one-GPU box needs to run the real one-process-per-rank control flow
 * of a shard group (unique-id hand-off over torch.distributed, RcclComm with device buffers, the same collective sequence in every process)
 * — real librccl refuses two ranks on one GPU.
 *
 *   rendezvous  a POSIX shared-memory segment named by the unique id (created by ncclGetUniqueId): a barrier, the ranks' op queues, pids.
 *   data path   device blocks travel by hipIpcGetMemHandle / hipIpcOpenMemHandle + a device-to-device copy on the receiver's stream;
 *               blocks that are not plain device allocations (pinned host aliases) and the small reduce go through a per-rank staging
 *               area inside the segment (hipMemcpy on both ends).
 *   semantics   SYNCHRONOUS: a collective first waits for the caller's stream, and returns after every rank has consumed the data. (RCCL
 *               enqueues a kernel and returns; the values that land in the buffers are the same.) Grouped point-to-point calls are matched
 *               per (sender, receiver) pair in issue order, as in the RCCL documentation.
 *   failures    every wait is bounded by BFHIP_COMM_TIMEOUT_S (the library's own setting) and watches the peers' pids: a rank that died makes
 *               the others return ncclInternalError (libbfhip: an error from the proof) instead of hanging.
 * Test infrastructure only: built by __graft_entry__.build() / tests with g++ against libamdhip64 into tests/libmock_rccl_ipc.so. */
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cerrno>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <map>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
enum { MAX_RANKS = 16, MAX_OPS = 4096 };
constexpr size_t STAGE_BYTES = size_t(32) << 20;      // per rank; only touched pages are ever backed by memory
constexpr uint32_t MAGIC = 0x6d6f636bu;

struct Block {                    // how a receiver finds the bytes of one published block
    int staged;                   // 1: at `off` inside the owner's staging area; 0: device memory behind an IPC handle
    hipIpcMemHandle_t handle;
    size_t off, bytes;
};
struct Op { int is_send, peer, used; Block b; void* recv_ptr; };
struct Shared {
    uint32_t magic; int n;
    std::atomic<int> joined, left, failed;
    std::atomic<uint32_t> bar_count, bar_gen;
    int pids[MAX_RANKS];
    int n_ops[MAX_RANKS];
    size_t stage_used[MAX_RANKS];
    Block coll[MAX_RANKS];
    Op ops[MAX_RANKS][MAX_OPS];
};
constexpr size_t HEADER_BYTES = (sizeof(Shared) + 4095) & ~size_t(4095);
constexpr size_t SEGMENT_BYTES = HEADER_BYTES + MAX_RANKS * STAGE_BYTES;

struct Comm {
    Shared* sh = nullptr; char* base = nullptr; int rank = 0; std::string name;
    std::map<std::string, void*> opened;          // peers' allocations mapped into this process, by handle bytes
    std::map<void*, hipIpcMemHandle_t> exported;  // my allocations, by base pointer
    char* stage(int r) const { return base + HEADER_BYTES + size_t(r) * STAGE_BYTES; }
};

double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
double timeout_s() { const char* v = getenv("BFHIP_COMM_TIMEOUT_S"); double x = v ? atof(v) : 0.0; return x > 0.0 ? x : 300.0; }
bool alive(int pid) {
    if (pid <= 0) return true;
    if (kill(pid, 0) != 0 && errno == ESRCH) return false;
    char path[64], buf[256];
    snprintf(path, sizeof path, "/proc/%d/stat", pid);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    size_t k = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[k] = 0;
    const char* p = strrchr(buf, ')');             // "pid (comm) S ..."
    return !(p && (p[2] == 'Z' || p[2] == 'X'));
}
// central barrier over the processes; -1: the group failed (a peer died, a peer gave up, or the wait ran out)
int barrier(Comm* c) {
    Shared* s = c->sh;
    if (s->failed.load()) return -1;
    const uint32_t gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == (uint32_t)s->n) { s->bar_count.store(0); s->bar_gen.fetch_add(1); return 0; }
    const double t0 = now(), limit = timeout_s();
    double next_check = 0.05;
    for (uint32_t polls = 1; s->bar_gen.load() == gen; polls++) {
        if (s->failed.load()) return -1;
        if (polls < 2000) continue;
        usleep(50);
        const double waited = now() - t0;
        if (waited > next_check) {
            next_check = waited + 0.05;
            for (int r = 0; r < s->n; r++) if (r != c->rank && !alive(s->pids[r])) { s->failed.store(1); return -1; }
            if (waited > limit) { s->failed.store(1); return -1; }
        }
    }
    return 0;
}
#define HIPOK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "mock_rccl_ipc: %s -> %s\n", #expr, hipGetErrorString(e__)); return ncclUnhandledCudaError; } } while (0)

// publishes one of MY blocks: device allocation -> IPC handle + offset; anything else -> copied into my staging area now
ncclResult_t publish(Comm* c, const void* ptr, size_t bytes, Block* out) {
    memset(out, 0, sizeof *out);
    out->bytes = bytes;
    if (!bytes) { out->staged = 1; return ncclSuccess; }
    hipPointerAttribute_t at;
    bool device = hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (device && !getenv("BFHIP_MOCK_RCCL_STAGE_ALL")) {
        void* b = nullptr; size_t sz = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&b, &sz, (hipDeviceptr_t)ptr) == hipSuccess && b) {
            auto it = c->exported.find(b);
            if (it == c->exported.end()) {
                hipIpcMemHandle_t h;
                if (hipIpcGetMemHandle(&h, b) == hipSuccess) it = c->exported.emplace(b, h).first;
                else (void)hipGetLastError();
            }
            if (it != c->exported.end()) { out->handle = it->second; out->off = (const char*)ptr - (const char*)b; return ncclSuccess; }
        }
        (void)hipGetLastError();
    }
    size_t& used = c->sh->stage_used[c->rank];
    const size_t at_off = (used + 255) & ~size_t(255);
    if (at_off + bytes > STAGE_BYTES) { fprintf(stderr, "mock_rccl_ipc: staging area exhausted (%zu bytes)\n", bytes); return ncclInternalError; }
    HIPOK(hipMemcpy(c->stage(c->rank) + at_off, ptr, bytes, hipMemcpyDefault));
    out->staged = 1; out->off = at_off; used = at_off + bytes;
    return ncclSuccess;
}
// copies a block published by rank `owner` to dst (stream-ordered for device blocks; the caller synchronises)
ncclResult_t fetch(Comm* c, int owner, const Block& b, void* dst, size_t skip, size_t bytes, hipStream_t s) {
    if (!bytes) return ncclSuccess;
    if (b.staged) { HIPOK(hipMemcpyAsync(dst, c->stage(owner) + b.off + skip, bytes, hipMemcpyDefault, s)); return ncclSuccess; }
    std::string key((const char*)&b.handle, sizeof b.handle);
    auto it = c->opened.find(key);
    if (it == c->opened.end()) {
        void* p = nullptr;
        HIPOK(hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess));
        it = c->opened.emplace(key, p).first;
    }
    HIPOK(hipMemcpyAsync(dst, (const char*)it->second + b.off + skip, bytes, hipMemcpyDeviceToDevice, s));
    return ncclSuccess;
}

thread_local int t_in_group = 0;
thread_local Comm* t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;
std::atomic<int> g_next_id{1};
}  // namespace

typedef Comm* ncclComm_t;

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/bfhip_mock_%d_%d", (int)getpid(), g_next_id.fetch_add(1));
    int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)SEGMENT_BYTES) != 0) { close(fd); shm_unlink(id->internal); return ncclSystemError; }
    void* m = mmap(nullptr, HEADER_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { shm_unlink(id->internal); return ncclSystemError; }
    Shared* s = (Shared*)m;               // a fresh segment is zero filled: counters, queues and flags start at 0
    s->magic = MAGIC;
    munmap(m, HEADER_BYTES);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
    if (n < 1 || n > MAX_RANKS || rank < 0 || rank >= n) return ncclInvalidArgument;
    id.internal[127] = 0;
    if (strncmp(id.internal, "/bfhip_mock_", 12) != 0) return ncclInvalidArgument;
    int fd = -1;
    for (double t0 = now(); fd < 0; ) { fd = shm_open(id.internal, O_RDWR, 0600); if (fd < 0) { if (now() - t0 > 30.0) return ncclSystemError; usleep(1000); } }
    void* m = mmap(nullptr, SEGMENT_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    Shared* s = (Shared*)m;
    if (s->magic != MAGIC) { munmap(m, SEGMENT_BYTES); return ncclInvalidArgument; }
    Comm* c = new Comm;
    c->sh = s; c->base = (char*)m; c->rank = rank; c->name = id.internal;
    if (rank == 0) s->n = n;
    s->pids[rank] = (int)getpid();
    s->joined.fetch_add(1);
    // like the real call: returns once every rank has joined (bounded: a rank that never shows up fails the group)
    for (double t0 = now(); s->joined.load() < n; usleep(200)) if (now() - t0 > timeout_s() || s->failed.load()) { s->failed.store(1); munmap(m, SEGMENT_BYTES); delete c; return ncclInternalError; }
    while (s->n == 0) usleep(50);
    if (s->n != n) { s->failed.store(1); munmap(m, SEGMENT_BYTES); delete c; return ncclInvalidArgument; }
    *out = c;
    return barrier(c) == 0 ? ncclSuccess : ncclInternalError;
}

static void release(Comm* c) {
    for (auto& kv : c->opened) (void)hipIpcCloseMemHandle(kv.second);
    (void)hipGetLastError();
    if (c->sh->left.fetch_add(1) + 1 >= c->sh->n) shm_unlink(c->name.c_str());      // the last one out removes the name
    munmap(c->base, SEGMENT_BYTES);
    delete c;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) { (void)barrier(c); release(c); } return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t c) { if (c) { c->sh->failed.store(1); release(c); } return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t* st) { *st = (c && c->sh->failed.load()) ? ncclInternalError : ncclSuccess; return ncclSuccess; }

ncclResult_t ncclGroupStart(void) { t_in_group = 1; t_comm = nullptr; t_stream = nullptr; return ncclSuccess; }
static ncclResult_t queue_op(int is_send, void* ptr, size_t count, int dt, int peer, ncclComm_t c, hipStream_t stream) {
    if (!t_in_group) return ncclInvalidUsage;     /* libbfhip always groups its point-to-point calls */
    if (dt != 1 /* ncclUint8 */ || peer < 0 || peer >= c->sh->n || peer == c->rank) return ncclInvalidArgument;
    Shared* s = c->sh;
    if (!t_comm) { HIPOK(hipStreamSynchronize(stream)); s->stage_used[c->rank] = 0; }     // first op of the group: my buffers are final
    if (s->n_ops[c->rank] == MAX_OPS) return ncclInternalError;
    Op& op = s->ops[c->rank][s->n_ops[c->rank]];
    memset(&op, 0, sizeof op);
    op.is_send = is_send; op.peer = peer; op.recv_ptr = ptr; op.b.bytes = count;
    if (is_send) { ncclResult_t r = publish(c, ptr, count, &op.b); if (r != ncclSuccess) return r; }
    s->n_ops[c->rank]++;
    t_comm = c; t_stream = stream;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* p, size_t count, int dt, int peer, ncclComm_t c, hipStream_t stream) { return queue_op(1, (void*)p, count, dt, peer, c, stream); }
ncclResult_t ncclRecv(void* p, size_t count, int dt, int peer, ncclComm_t c, hipStream_t stream) { return queue_op(0, p, count, dt, peer, c, stream); }
ncclResult_t ncclGroupEnd(void) {
    t_in_group = 0;
    Comm* c = t_comm;
    if (!c) return ncclSuccess;                   /* an empty group: nothing was queued on this rank (all ranks must agree) */
    Shared* s = c->sh;
    ncclResult_t rc = ncclSuccess;
    if (barrier(c) != 0) return ncclInternalError;          /* every rank has posted its queue */
    for (int k = 0; k < s->n_ops[c->rank]; k++) {
        Op* r = &s->ops[c->rank][k];
        if (r->is_send) continue;
        Op* snd = nullptr;
        for (int j = 0; j < s->n_ops[r->peer] && !snd; j++) {
            Op* q = &s->ops[r->peer][j];
            if (q->is_send && q->peer == c->rank && !q->used) snd = q;
        }
        if (!snd || snd->b.bytes != r->b.bytes) { rc = ncclInvalidUsage; if (snd) snd->used = 1; continue; }
        snd->used = 1;
        ncclResult_t f = fetch(c, r->peer, snd->b, r->recv_ptr, 0, r->b.bytes, t_stream);
        if (f != ncclSuccess) rc = f;
    }
    if (hipStreamSynchronize(t_stream) != hipSuccess) rc = ncclUnhandledCudaError;
    if (barrier(c) != 0) return ncclInternalError;          /* every rank has consumed what was sent to it */
    for (int k = 0; k < s->n_ops[c->rank]; k++) if (s->ops[c->rank][k].is_send && !s->ops[c->rank][k].used) rc = ncclInvalidUsage;   /* a send nobody received */
    s->n_ops[c->rank] = 0;
    if (barrier(c) != 0) return ncclInternalError;
    return rc;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int dt, ncclComm_t c, hipStream_t stream) {
    if (dt != 1) return ncclInvalidArgument;
    Shared* s = c->sh;
    HIPOK(hipStreamSynchronize(stream));
    s->stage_used[c->rank] = 0;
    ncclResult_t rc = publish(c, send, count, &s->coll[c->rank]);
    if (rc != ncclSuccess) { s->failed.store(1); return rc; }
    if (barrier(c) != 0) return ncclInternalError;
    for (int r = 0; r < s->n; r++) {
        char* dst = (char*)recv + (size_t)r * count;
        if (r == c->rank && dst == (const char*)send) continue;       /* in place: my block already lies where it belongs */
        ncclResult_t f = fetch(c, r, s->coll[r], dst, 0, count, stream);
        if (f != ncclSuccess) rc = f;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) rc = ncclUnhandledCudaError;
    if (barrier(c) != 0) return ncclInternalError;
    return rc;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, ncclComm_t c, hipStream_t stream) {
    if (dt != 3 /* ncclUint32 */ || op != 2 /* ncclMax */) return ncclInvalidArgument;
    Shared* s = c->sh;
    const size_t bytes = count * 4;
    if (bytes > STAGE_BYTES) return ncclInternalError;
    HIPOK(hipStreamSynchronize(stream));
    HIPOK(hipMemcpy(c->stage(c->rank), send, bytes, hipMemcpyDefault));      /* the reduce is taken on the host over the ranks' staged inputs */
    if (barrier(c) != 0) return ncclInternalError;
    std::vector<uint32_t> out(count ? count : 1, 0u);
    for (int r = 0; r < s->n; r++) { const uint32_t* v = (const uint32_t*)c->stage(r); for (size_t i = 0; i < count; i++) if (v[i] > out[i]) out[i] = v[i]; }
    HIPOK(hipMemcpy(recv, out.data(), bytes, hipMemcpyDefault));
    if (barrier(c) != 0) return ncclInternalError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (mock: unmatched send/receive)" : r == ncclInvalidArgument ? "invalid argument"
         : r == ncclUnhandledCudaError ? "HIP error (mock)" : r == ncclSystemError ? "system error (mock: shared memory)" : "internal error (mock: a peer died, gave up, or the wait ran out)";
}

}  // extern "C"
