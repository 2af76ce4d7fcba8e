/* Test double of the 10 RCCL entry points libbfhip binds with dlsym (stwo-brainfuck_amd/csrc/comm.hip: RcclApi), selected with the
 * environment variable BFHIP_RCCL_LIBRARY. The "ranks" are THREADS of one process and every buffer is HOST memory, so RcclComm's
 * bookkeeping — blocks to oneself, zero-byte blocks, several blocks per peer, matching order, size mismatches — runs on a box without a GPU
 * (real ncclSend/ncclRecv need one GPU per rank). Semantics follow the RCCL documentation: point-to-point calls between ncclGroupStart and
 * ncclGroupEnd are matched per (sender, receiver) pair in issue order. Test infrastructure only; built by tests/test_abi_and_replicas.py. */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
typedef struct { char internal[128]; } ncclUniqueId;
enum { MAX_RANKS = 16, MAX_OPS = 256, MAX_GROUPS = 64 };

typedef struct { int is_send, peer; void* ptr; size_t bytes; int used; } Op;
typedef struct Group {
    char id[128]; int n, joined;
    pthread_mutex_t mu; pthread_cond_t cv; int arrived; unsigned long gen;
    Op ops[MAX_RANKS][MAX_OPS]; int n_ops[MAX_RANKS];
    const void* coll_src[MAX_RANKS];
} Group;
typedef struct { Group* g; int rank; } Comm;
typedef Comm* ncclComm_t;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static Group* g_groups[MAX_GROUPS];
static int g_n_groups, g_next_id = 1;
static __thread int t_in_group;
static __thread Comm* t_comm;

static void barrier(Group* g) {
    pthread_mutex_lock(&g->mu);
    unsigned long gen = g->gen;
    if (++g->arrived == g->n) { g->arrived = 0; g->gen++; pthread_cond_broadcast(&g->cv); }
    else while (g->gen == gen) pthread_cond_wait(&g->cv, &g->mu);
    pthread_mutex_unlock(&g->mu);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    pthread_mutex_lock(&g_mu);
    int v = g_next_id++;
    pthread_mutex_unlock(&g_mu);
    memcpy(id->internal, "mock", 4); memcpy(id->internal + 4, &v, sizeof v);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
    if (n < 1 || n > MAX_RANKS || rank < 0 || rank >= n) return ncclInvalidArgument;
    pthread_mutex_lock(&g_mu);
    Group* g = NULL;
    for (int i = 0; i < g_n_groups; i++) if (memcmp(g_groups[i]->id, id.internal, 128) == 0) g = g_groups[i];
    if (!g) {
        if (g_n_groups == MAX_GROUPS) { pthread_mutex_unlock(&g_mu); return ncclInternalError; }
        g = (Group*)calloc(1, sizeof(Group));
        memcpy(g->id, id.internal, 128); g->n = n;
        pthread_mutex_init(&g->mu, NULL); pthread_cond_init(&g->cv, NULL);
        g_groups[g_n_groups++] = g;
    }
    g->joined++;
    pthread_mutex_unlock(&g_mu);
    if (g->n != n) return ncclInvalidArgument;
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    c->g = g; c->rank = rank;
    *out = c;
    barrier(g);                                   /* like the real call: returns once every rank has joined */
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) { barrier(c->g); free(c); } return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { t_in_group = 1; t_comm = NULL; return ncclSuccess; }
static ncclResult_t queue(int is_send, void* ptr, size_t count, int dt, int peer, ncclComm_t c) {
    if (!t_in_group) return ncclInvalidUsage;     /* libbfhip always groups its point-to-point calls */
    if (dt != 1 /* ncclUint8 */ || peer < 0 || peer >= c->g->n || peer == c->rank) return ncclInvalidArgument;
    Group* g = c->g;
    if (g->n_ops[c->rank] == MAX_OPS) return ncclInternalError;
    Op op = {is_send, peer, ptr, count, 0};
    g->ops[c->rank][g->n_ops[c->rank]++] = op;
    t_comm = c;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* p, size_t count, int dt, int peer, ncclComm_t c, void* stream) { (void)stream; return queue(1, (void*)p, count, dt, peer, c); }
ncclResult_t ncclRecv(void* p, size_t count, int dt, int peer, ncclComm_t c, void* stream) { (void)stream; return queue(0, p, count, dt, peer, c); }
ncclResult_t ncclGroupEnd(void) {
    t_in_group = 0;
    Comm* c = t_comm;
    if (!c) return ncclSuccess;                   /* an empty group: nothing was queued on this rank (all ranks must agree) */
    Group* g = c->g;
    ncclResult_t rc = ncclSuccess;
    barrier(g);                                   /* every rank has posted its queue */
    for (int k = 0; k < g->n_ops[c->rank]; k++) {
        Op* r = &g->ops[c->rank][k];
        if (r->is_send) continue;
        Op* s = NULL;
        for (int j = 0; j < g->n_ops[r->peer] && !s; j++) {
            Op* q = &g->ops[r->peer][j];
            if (q->is_send && q->peer == c->rank && !q->used) s = q;
        }
        if (!s || s->bytes != r->bytes) { rc = ncclInvalidUsage; if (s) s->used = 1; continue; }
        s->used = 1;
        memcpy(r->ptr, s->ptr, r->bytes);
    }
    barrier(g);                                   /* every rank has consumed what was sent to it */
    for (int k = 0; k < g->n_ops[c->rank]; k++) if (g->ops[c->rank][k].is_send && !g->ops[c->rank][k].used) rc = ncclInvalidUsage;   /* a send nobody received */
    g->n_ops[c->rank] = 0;
    barrier(g);
    return rc;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int dt, ncclComm_t c, void* stream) {
    (void)stream;
    if (dt != 1) return ncclInvalidArgument;
    Group* g = c->g;
    char* tmp = (char*)malloc(count ? count : 1);
    memcpy(tmp, send, count);                     /* in place: the send block lies inside recv */
    g->coll_src[c->rank] = tmp;
    barrier(g);
    for (int r = 0; r < g->n; r++) memcpy((char*)recv + (size_t)r * count, g->coll_src[r], count);
    barrier(g);
    free(tmp);
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, int dt, int op, ncclComm_t c, void* stream) {
    (void)stream;
    if (dt != 3 /* ncclUint32 */ || op != 2 /* ncclMax */) return ncclInvalidArgument;
    Group* g = c->g;
    uint32_t* tmp = (uint32_t*)malloc(count * 4 + 4);
    memcpy(tmp, send, count * 4);
    g->coll_src[c->rank] = tmp;
    barrier(g);
    for (size_t i = 0; i < count; i++) { uint32_t m = 0; for (int r = 0; r < g->n; r++) { uint32_t v = ((const uint32_t*)g->coll_src[r])[i]; if (v > m) m = v; } ((uint32_t*)recv)[i] = m; }
    barrier(g);
    free(tmp);
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (mock: unmatched send/receive)" : r == ncclInvalidArgument ? "invalid argument" : "internal error"; }
int bfhip_mock_self_copy(void* dst, const void* src, size_t bytes) { memmove(dst, src, bytes); return 0; }
