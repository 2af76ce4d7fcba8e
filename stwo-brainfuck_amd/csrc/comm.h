// Shard-group communication: the exchanges of ONE proof over several GPUs, issued by the C++ prover on the
// context's own HIP stream with device buffers on both ends — no host callbacks and no host bounce in the data path.
//
// Two transports behind one interface:
//   RcclComm   one process per GPU (bench.py --gpus N --shard): RCCL collectives / grouped send-recv over xGMI. librccl is loaded with
//              dlopen at group-join time, so the library itself has no link-time dependency on it.
//   LocalComm  N contexts of ONE process driven by N host threads (tests: N ranks on one GPU): device-to-device copies ordered by HIP
//              events between the ranks' streams plus a host rendezvous. Same call sequence, same bytes.
// Every rank of a group calls the same sequence of operations with matching sizes (SPMD), like any collective library expects.
#pragma once
#include "kernels.h"
#include <vector>
#include <memory>
#include <cstring>

namespace bf {

struct Xfer { u32 peer; void* ptr; size_t bytes; };   // one point-to-point block: send (ptr is the source) or receive (ptr is the destination)

struct Comm {
    u32 rank = 0, count = 1;
    // operation counts and payload bytes this rank sent to OTHER ranks since the group was joined (tests, DESIGN.md numbers)
    u64 n_all_gather = 0, n_all_reduce = 0, n_exchange = 0, bytes_sent = 0;
    virtual ~Comm();
    // In place: rank r's block is buf + r * bytes_per_rank; afterwards every block is filled on every rank.
    virtual void all_gather(hipStream_t s, void* buf, size_t bytes_per_rank) = 0;
    // Element-wise maximum over the ranks (decommitment words / sampled values: each is held by one rank, zero elsewhere).
    virtual void all_reduce_max_u32(hipStream_t s, u32* buf, size_t n) = 0;
    // Point-to-point blocks. Sends to / receives from one peer are matched in list order. A block to oneself is a plain copy.
    virtual void exchange(hipStream_t s, const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs) = 0;
    virtual const char* transport() const = 0;
    // true when the ranks of the group sit on more than one GPU (RCCL: always; in-process: known once the first collective has seen every
    // member's device) — then an exchange is a transfer over xGMI that the prover overlaps with the transforms still to be done
    virtual bool spans_devices() const { return false; }
    // Asynchronous failure of the transport (RCCL: ncclCommGetAsyncError). Called while the host waits for the stream; throws.
    virtual void check_async() {}
    // Gives up on the group after a failure or a timeout so that blocked peers and streams are released (RCCL: ncclCommAbort).
    virtual void abort() {}

    // GPU-side duration of every collective (one event pair each, on the stream that carries it): where the time of a proof over several
    // GPUs goes — [0] all-gathers, [1] max-reduces, [2] grouped send-receives, in milliseconds since the group was joined. The stream must
    // have been synchronised. Not recorded for a null stream (host-memory test transport).
    enum { T_ALL_GATHER = 0, T_ALL_REDUCE = 1, T_EXCHANGE = 2 };
    void times_ms(double out[3]);
    // Per-collective latency since the join (or the last reset): for each kind {count, GPU-side p50 / p90 / max, host-side p50 / p90 / max} in
    // microseconds — GPU side = the event pair around the collective on its stream (includes waiting for the slowest peer), host side = the time the
    // calling thread spent inside the transport's call (RCCL: the enqueue). A proof over N GPUs issues ~31 collectives: their LATENCY, not their bytes,
    // is what the first hardware run has to show. The stream must have been synchronised. reset: forget the history afterwards.
    void latency_us(double out[3][7], bool reset);
  protected:
    struct TimedOp { hipEvent_t e0, e1; int kind; };
    static constexpr size_t MAX_LATENCIES = 8192;        // per kind; older entries are dropped
    std::vector<float> gpu_us_[3], host_us_[3];
    std::vector<TimedOp> pending_;
    std::vector<hipEvent_t> pool_;
    double ms_[3] = {0, 0, 0};
    hipEvent_t timing_event();
    void drain_completed();
    static constexpr size_t MAX_PENDING = 512;
    struct Timed {      // brackets one collective
        Comm& c; hipStream_t s; hipEvent_t e1 = nullptr; int kind; double t0;
        Timed(Comm& c_, hipStream_t s_, int kind);
        ~Timed();
    };
};

// LocalComm rendezvous object shared by the N contexts of one process (reference counted: it lives until the last member has left).
struct LocalGroup;
std::shared_ptr<LocalGroup> local_group_create(u32 count);
std::unique_ptr<Comm> local_comm_join(const std::shared_ptr<LocalGroup>& g, u32 rank);

// RCCL: rank 0 creates the 128-byte unique id, the host program distributes it (control plane), every rank joins.
void rccl_unique_id(unsigned char id[128]);
std::unique_ptr<Comm> rccl_comm_join(const unsigned char id[128], u32 rank, u32 count);
// Seconds a host wait inside a shard group may last before the group is declared failed (BFHIP_COMM_TIMEOUT_S, default 300).
double comm_timeout_seconds();

}  // namespace bf
