/* A stand-in for librccl for tests/test_comm_stub.py (CPU): the eight entry points lle_amd/csrc/comm.cpp binds with dlsym, over HOST memory.
 * One process owns every rank (ncclCommInitAll); all-reduces are collected between ncclGroupStart / ncclGroupEnd and carried out at the
 * end of the group, which is what lets one thread post the calls of all ranks.  Failure injection: FAKE_RCCL_FAIL = "initall" |
 * "allreduce<k>" (the k-th ncclAllReduce call of the process fails).  Counters for the test: fake_rccl_live_comms(), fake_rccl_group_depth(). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int rank, n; int id; } comm_t;
typedef struct { comm_t* c; const int64_t* send; int64_t* recv; size_t count; int op; } call_t;
static int g_live = 0, g_depth = 0, g_next_id = 1, g_calls = 0, g_n_pending = 0;
static call_t g_pending[128];

static int fail(const char* what) { const char* e = getenv("FAKE_RCCL_FAIL"); return e && !strcmp(e, what); }
int fake_rccl_live_comms(void) { return g_live; }
int fake_rccl_group_depth(void) { return g_depth; }

int ncclGetUniqueId(void* id) { memset(id, 7, 128); return 0; }
int ncclCommInitRank(void** comm, int n, char id[128], int rank) { (void)id; comm_t* c = calloc(1, sizeof *c); c->rank = rank; c->n = n; c->id = 0; *comm = c; g_live++; return 0; }
int ncclCommInitAll(void** comms, int n, const int* devs) {
    (void)devs;
    if (fail("initall")) return 1;
    const int id = g_next_id++;
    for (int k = 0; k < n; k++) { comm_t* c = calloc(1, sizeof *c); c->rank = k; c->n = n; c->id = id; comms[k] = c; g_live++; }
    return 0;
}
int ncclCommDestroy(void* comm) { free(comm); g_live--; return 0; }
int ncclGroupStart(void) { g_depth++; return 0; }
static void run_pending(void) {
    for (int i = 0; i < g_n_pending; i++) {            /* every call's result = the reduction over the calls of its communicator */
        call_t* a = &g_pending[i];
        for (size_t e = 0; e < a->count; e++) {
            int64_t acc = 0; int first = 1;
            for (int j = 0; j < g_n_pending; j++) {
                call_t* b = &g_pending[j];
                if (b->c->id != a->c->id) continue;
                const int64_t v = b->send[e];          /* (in place: read before any write below -- results go to a scratch first) */
                acc = first ? v : (a->op == 0 ? acc + v : (v > acc ? v : acc));
                first = 0;
            }
            ((int64_t*)a->recv)[e + a->count] = acc;   /* scratch: the test's buffers are 2 x count long */
        }
    }
    for (int i = 0; i < g_n_pending; i++) memcpy(g_pending[i].recv, g_pending[i].recv + g_pending[i].count, g_pending[i].count * 8);
    g_n_pending = 0;
}
int ncclGroupEnd(void) { if (--g_depth == 0) run_pending(); return 0; }
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
    (void)stream;
    char name[32];
    g_calls++;
    strcpy(name, "allreduce0"); name[9] = (char)('0' + g_calls);
    if (fail(name) || dtype != 4) return 2;
    g_pending[g_n_pending++] = (call_t){(comm_t*)comm, send, recv, count, op == 0 ? 0 : 1};
    if (g_depth == 0) run_pending();
    return 0;
}
const char* ncclGetErrorString(int r) { return r == 1 ? "fake: init failed" : "fake: all-reduce failed"; }
