/* Process-per-rank test double of the RCCL entry points libbfhip binds with dlsym (stwo-brainfuck_amd/csrc/comm.hip: RcclApi), selected with
 * the environment variable BFHIP_RCCL_LIBRARY. Unlike tests/mock_rccl.c (ranks = threads, host buffers) the ranks here are PROCESSES, each
 * with its own HIP context, and the buffers are DEVICE memory: what a one-GPU box needs to run the real one-process-per-rank control flow
 * of a shard group (unique-id hand-off over torch.distributed, RcclComm with device buffers, the same collective sequence in every process)
 * — real librccl refuses two ranks on one GPU.
 *
 *   rendezvous  a POSIX shared-memory segment named by the unique id (created by ncclGetUniqueId): a barrier, per-round manifests, pids.
 *   data path   every rank owns ONE device bounce buffer (64 MiB, hipMalloc), exported once with hipIpcGetMemHandle and opened by every peer at
 *               communicator creation. A transfer is: sender copies a piece of its block into its bounce buffer (device to device), rendezvous,
 *               receiver copies from the peer's mapped bounce buffer to its destination, rendezvous; blocks larger than the buffer take several
 *               rounds. (The caller's own allocations are never exported: mapping a 2 GiB arena chunk of another process did not return on
 *               this pool.) The small max-reduce goes through a host staging area inside the segment.
 *   semantics   SYNCHRONOUS: a collective first waits for the caller's stream, and returns after every rank has consumed the data. (RCCL
 *               enqueues a kernel and returns; the values that land in the buffers are the same.) Grouped point-to-point calls are matched
 *               per (sender, receiver) pair in issue order, as in the RCCL documentation.
 *   failures    every wait is bounded by BFHIP_COMM_TIMEOUT_S (the library's own setting) and watches the peers' pids: a rank that died makes
 *               the others return ncclInternalError (libbfhip: an error from the proof) instead of hanging.
 * BFHIP_MOCK_RCCL_TRACE=1: one stderr line per collective and round. Test infrastructure only: built by __graft_entry__.build() / tests with g++
 * against libamdhip64 into tests/libmock_rccl_ipc.so. */
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
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
enum { MAX_RANKS = 16, MAX_SEGS = 4096 };
constexpr size_t BOUNCE_BYTES = size_t(64) << 20;     // device memory per rank
constexpr size_t STAGE_BYTES = size_t(32) << 20;      // host staging per rank (max-reduce); only touched pages are ever backed by memory
constexpr uint32_t MAGIC = 0x6d6f636cu;

// one piece of one send of this round: bytes [block_off, block_off + len) of the sender's pair_seq-th block to `peer` lie at bounce_off
struct Seg { int peer, pair_seq; size_t block_bytes, block_off, len, bounce_off; };
struct Shared {
    uint32_t magic; int n;
    std::atomic<int> joined, left, failed;
    std::atomic<uint32_t> bar_count, bar_gen;
    int pids[MAX_RANKS];
    hipIpcMemHandle_t bounce[MAX_RANKS];
    int n_segs[MAX_RANKS], more[MAX_RANKS];
    int n_recv_from[MAX_RANKS][MAX_RANKS];            // [receiver][sender]: receives posted in the current group
    Seg segs[MAX_RANKS][MAX_SEGS];
};
constexpr size_t HEADER_BYTES = (sizeof(Shared) + 4095) & ~size_t(4095);
constexpr size_t SEGMENT_BYTES = HEADER_BYTES + MAX_RANKS * STAGE_BYTES;

struct Comm {
    Shared* sh = nullptr; char* base = nullptr; int rank = 0; std::string name;
    char* bounce = nullptr;                       // my device bounce buffer
    char* peer_bounce[MAX_RANKS] = {};            // the peers' bounce buffers mapped into this process
    char* stage(int r) const { return base + HEADER_BYTES + size_t(r) * STAGE_BYTES; }
};
struct LocalOp { int is_send, peer; char* ptr; size_t bytes, done; };

bool tracing() { static const bool on = getenv("BFHIP_MOCK_RCCL_TRACE") != nullptr; return on; }
#define TRACE(...) do { if (tracing()) { fprintf(stderr, "mock_rccl_ipc[%d] ", (int)getpid()); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)
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

thread_local int t_in_group = 0;
thread_local Comm* t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;
thread_local std::vector<LocalOp> t_ops;
std::atomic<int> g_next_id{1};

void release(Comm* c, bool orderly) {
    for (int r = 0; r < MAX_RANKS; r++) if (c->peer_bounce[r]) (void)hipIpcCloseMemHandle(c->peer_bounce[r]);
    if (orderly) (void)barrier(c);                 // every peer has unmapped my buffer before it is freed
    if (c->bounce) (void)hipFree(c->bounce);
    (void)hipGetLastError();
    if (c->sh->left.fetch_add(1) + 1 >= c->sh->n) shm_unlink(c->name.c_str());      // the last one out removes the name
    munmap(c->base, SEGMENT_BYTES);
    delete c;
}
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
    Shared* s = (Shared*)m;               // a fresh segment is zero filled: counters, manifests and flags start at 0
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
    if (hipMalloc((void**)&c->bounce, BOUNCE_BYTES) != hipSuccess || hipIpcGetMemHandle(&s->bounce[rank], c->bounce) != hipSuccess) {
        fprintf(stderr, "mock_rccl_ipc: bounce buffer: %s\n", hipGetErrorString(hipGetLastError()));
        s->failed.store(1); munmap(m, SEGMENT_BYTES); delete c; return ncclUnhandledCudaError;
    }
    if (rank == 0) s->n = n;
    s->pids[rank] = (int)getpid();
    s->joined.fetch_add(1);
    // like the real call: returns once every rank has joined (bounded: a rank that never shows up fails the group)
    for (double t0 = now(); s->joined.load() < n; usleep(200)) if (now() - t0 > timeout_s() || s->failed.load()) { s->failed.store(1); release(c, false); return ncclInternalError; }
    while (s->n == 0) usleep(50);
    if (s->n != n) { s->failed.store(1); release(c, false); return ncclInvalidArgument; }
    if (barrier(c) != 0) { release(c, false); return ncclInternalError; }
    for (int r = 0; r < n; r++) {
        if (r == rank) continue;
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, s->bounce[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            fprintf(stderr, "mock_rccl_ipc: hipIpcOpenMemHandle of rank %d: %s\n", r, hipGetErrorString(hipGetLastError()));
            s->failed.store(1); release(c, false); return ncclUnhandledCudaError;
        }
        c->peer_bounce[r] = (char*)p;
    }
    TRACE("rank %d of %d joined, bounce buffers mapped", rank, n);
    *out = c;
    return barrier(c) == 0 ? ncclSuccess : ncclInternalError;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) { (void)barrier(c); release(c, true); } return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t c) { if (c) { c->sh->failed.store(1); release(c, false); } return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t* st) { *st = (c && c->sh->failed.load()) ? ncclInternalError : ncclSuccess; return ncclSuccess; }

ncclResult_t ncclGroupStart(void) { t_in_group = 1; t_comm = nullptr; t_stream = nullptr; t_ops.clear(); return ncclSuccess; }
static ncclResult_t queue_op(int is_send, void* ptr, size_t count, int dt, int peer, ncclComm_t c, hipStream_t stream) {
    if (!t_in_group) return ncclInvalidUsage;     /* libbfhip always groups its point-to-point calls */
    if (dt != 1 /* ncclUint8 */ || peer < 0 || peer >= c->sh->n || peer == c->rank) return ncclInvalidArgument;
    if (t_comm && t_comm != c) return ncclInvalidUsage;
    t_ops.push_back(LocalOp{is_send, peer, (char*)ptr, count, 0});
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
    hipStream_t stream = t_stream;
    ncclResult_t rc = ncclSuccess;
    TRACE("group: %zu ops, waiting for the stream", t_ops.size());
    HIPOK(hipStreamSynchronize(stream));         /* my send buffers are final */
    std::vector<LocalOp*> sends, recvs[MAX_RANKS];
    std::vector<int> send_seq;
    int n_send_to[MAX_RANKS] = {};
    for (auto& op : t_ops) {
        if (op.is_send) { sends.push_back(&op); send_seq.push_back(n_send_to[op.peer]++); }
        else recvs[op.peer].push_back(&op);
    }
    for (int p = 0; p < s->n; p++) s->n_recv_from[c->rank][p] = (int)recvs[p].size();
    size_t i = 0, off = 0;
    for (int round = 0;; round++) {
        /* pack as much of my pending sends as fits into my bounce buffer */
        size_t used = 0; int n_segs = 0;
        while (i < sends.size() && n_segs < MAX_SEGS) {
            const size_t at = (used + 255) & ~size_t(255);
            if (at >= BOUNCE_BYTES) break;
            const size_t len = sends[i]->bytes - off < BOUNCE_BYTES - at ? sends[i]->bytes - off : BOUNCE_BYTES - at;
            if (len) HIPOK(hipMemcpyAsync(c->bounce + at, sends[i]->ptr + off, len, hipMemcpyDefault, stream));
            s->segs[c->rank][n_segs++] = Seg{sends[i]->peer, send_seq[i], sends[i]->bytes, off, len, at};
            used = at + len; off += len;
            if (off == sends[i]->bytes) { i++; off = 0; }
        }
        HIPOK(hipStreamSynchronize(stream));
        s->n_segs[c->rank] = n_segs;
        s->more[c->rank] = i < sends.size() ? 1 : 0;
        if (barrier(c) != 0) return ncclInternalError;      /* every rank has published its round */
        if (round == 0)
            for (int p = 0; p < s->n; p++) if (p != c->rank && s->n_recv_from[p][c->rank] != n_send_to[p]) rc = ncclInvalidUsage;      /* a send nobody receives / a receive nobody sends */
        for (int p = 0; p < s->n; p++) {
            if (p == c->rank) continue;
            for (int k = 0; k < s->n_segs[p]; k++) {
                const Seg& g = s->segs[p][k];
                if (g.peer != c->rank) continue;
                if (g.pair_seq >= (int)recvs[p].size() || recvs[p][g.pair_seq]->bytes != g.block_bytes) { rc = ncclInvalidUsage; continue; }
                LocalOp* r = recvs[p][g.pair_seq];
                if (g.len) HIPOK(hipMemcpyAsync(r->ptr + g.block_off, c->peer_bounce[p] + g.bounce_off, g.len, hipMemcpyDeviceToDevice, stream));
                r->done += g.len;
            }
        }
        HIPOK(hipStreamSynchronize(stream));
        int any_more = 0;
        for (int p = 0; p < s->n; p++) any_more |= s->more[p];
        TRACE("group round %d: sent %zu bytes in %d pieces, more rounds: %d", round, used, n_segs, any_more);
        if (barrier(c) != 0) return ncclInternalError;      /* every rank has consumed the round: the bounce buffers may be overwritten */
        if (!any_more) break;
    }
    for (int p = 0; p < s->n; p++) for (auto* r : recvs[p]) if (r->done != r->bytes) rc = ncclInvalidUsage;
    t_ops.clear();
    return rc;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int dt, ncclComm_t c, hipStream_t stream) {
    if (dt != 1) return ncclInvalidArgument;
    Shared* s = c->sh;
    TRACE("all-gather %zu bytes per rank", count);
    HIPOK(hipStreamSynchronize(stream));
    char* mine = (char*)recv + (size_t)c->rank * count;
    if (mine != (const char*)send && count) HIPOK(hipMemcpyAsync(mine, send, count, hipMemcpyDefault, stream));      /* not in place */
    for (size_t off = 0; off < count || off == 0; off += BOUNCE_BYTES) {
        const size_t len = count - off < BOUNCE_BYTES ? count - off : BOUNCE_BYTES;
        if (len) HIPOK(hipMemcpyAsync(c->bounce, (const char*)send + off, len, hipMemcpyDefault, stream));
        HIPOK(hipStreamSynchronize(stream));
        if (barrier(c) != 0) return ncclInternalError;
        for (int r = 0; r < s->n; r++)
            if (r != c->rank && len) HIPOK(hipMemcpyAsync((char*)recv + (size_t)r * count + off, c->peer_bounce[r], len, hipMemcpyDeviceToDevice, stream));
        HIPOK(hipStreamSynchronize(stream));
        if (barrier(c) != 0) return ncclInternalError;
        if (count == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, ncclComm_t c, hipStream_t stream) {
    if (dt != 3 /* ncclUint32 */ || op != 2 /* ncclMax */) return ncclInvalidArgument;
    Shared* s = c->sh;
    const size_t bytes = count * 4;
    if (bytes > STAGE_BYTES) return ncclInternalError;
    TRACE("max-reduce %zu words", count);
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
