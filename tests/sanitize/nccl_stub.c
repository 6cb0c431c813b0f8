// TEST-ONLY stand-in for the five RCCL entry points the C host layer binds at run time (host/pbr_gather.c): grouped
// point-to-point transfers between the PROCESSES of a CPU test, over AF_UNIX socket pairs, with RCCL's semantics for what the
// exchange relies on: calls between ncclGroupStart and ncclGroupEnd are only queued; ncclGroupEnd progresses all of them
// together (no ordering between peers, FIFO per peer and direction); a send matches the peer's next receive from this rank
// and their byte counts must agree.  "Device memory" is host memory and the "stream" is ignored (the fake backend of
// gather_world.c executes kernels at submit time, so stream order == program order).
// Built as libnccl_stub.so by tests/test_host_cpu.py and selected with PBR_SetRcclLibrary(); never part of the product.
#define _GNU_SOURCE 1
#include <errno.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define STUB_MAX_WORLD 16
#define STUB_MAX_OPS 8192

typedef struct StubComm { int world, rank; int fd[STUB_MAX_WORLD]; } StubComm;
typedef struct StubOp { int send; char* ptr; size_t bytes; int peer; StubComm* comm; size_t done; int header_done; } StubOp;

static StubOp g_ops[STUB_MAX_OPS];
static int g_nops, g_depth;
static long g_starts, g_ends, g_sends, g_recvs;
static long g_fail_send_at = -1, g_fail_recv_at = -1;      /* fail the k-th call (0-based) with ncclInternalError (3) */

/* ---- test controls (not RCCL API) ---- */
void* stub_comm_create(int world, int rank, const int* fds) {
    if (world < 1 || world > STUB_MAX_WORLD) return NULL;
    StubComm* c = (StubComm*)calloc(1, sizeof *c);
    c->world = world; c->rank = rank;
    for (int i = 0; i < world; ++i) c->fd[i] = fds ? fds[i] : -1;
    return c;
}
void stub_comm_destroy(void* c) { free(c); }
void stub_fail_send_at(long k) { g_fail_send_at = k; }
void stub_fail_recv_at(long k) { g_fail_recv_at = k; }
void stub_counters(long* starts, long* ends, long* sends, long* recvs, int* depth) {
    if (starts) *starts = g_starts; if (ends) *ends = g_ends; if (sends) *sends = g_sends; if (recvs) *recvs = g_recvs; if (depth) *depth = g_depth;
}

/* ---- the RCCL entry points ---- */
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 3 ? "internal error (stub)" : r == 4 ? "invalid argument (stub)" : "error (stub)"; }
int ncclGetVersion(int* v) { *v = 0; return 0; }
int ncclCommCount(void* c, int* n) { *n = ((StubComm*)c)->world; return 0; }
int ncclCommUserRank(void* c, int* r) { *r = ((StubComm*)c)->rank; return 0; }
int ncclGroupStart(void) { ++g_starts; ++g_depth; return 0; }

static int queue_op(int send, const void* p, size_t bytes, int peer, void* comm) {
    StubComm* c = (StubComm*)comm;
    if (!c || peer < 0 || peer >= c->world || !p) return 4;
    if (g_nops >= STUB_MAX_OPS) return 3;
    StubOp* o = &g_ops[g_nops++];
    o->send = send; o->ptr = (char*)(uintptr_t)p; o->bytes = bytes; o->peer = peer; o->comm = c; o->done = 0; o->header_done = 0;
    return 0;
}

static int progress_all(void);

int ncclSend(const void* p, size_t count, int dtype, int peer, void* comm, void* stream) {
    (void)stream;
    if (dtype != 0) return 4;                               /* the exchange moves bytes (ncclInt8) */
    if (g_sends++ == g_fail_send_at) return 3;
    int r = queue_op(1, p, count, peer, comm);
    if (r == 0 && g_depth == 0) r = progress_all();
    return r;
}
int ncclRecv(void* p, size_t count, int dtype, int peer, void* comm, void* stream) {
    (void)stream;
    if (dtype != 0) return 4;
    if (g_recvs++ == g_fail_recv_at) return 3;
    int r = queue_op(0, p, count, peer, comm);
    if (r == 0 && g_depth == 0) r = progress_all();
    return r;
}

/* the head-of-line op of (peer, direction): FIFO per peer and direction */
static int is_head(int k) {
    for (int j = 0; j < k; ++j)
        if (g_ops[j].comm == g_ops[k].comm && g_ops[j].peer == g_ops[k].peer && g_ops[j].send == g_ops[k].send && g_ops[j].done < g_ops[j].bytes + 1) return 0;
    return 1;
}

static int progress_all(void) {
    /* self transfers: k-th send to self pairs with the k-th receive from self */
    for (int i = 0; i < g_nops; ++i) {
        StubOp* s = &g_ops[i];
        if (!s->send || s->peer != s->comm->rank || s->done) continue;
        int found = 0;
        for (int j = 0; j < g_nops; ++j) {
            StubOp* r = &g_ops[j];
            if (r->send || r->comm != s->comm || r->peer != r->comm->rank || r->done) continue;
            if (r->bytes != s->bytes) { fprintf(stderr, "nccl stub: self send of %zu bytes meets a receive of %zu\n", s->bytes, r->bytes); g_nops = 0; return 3; }
            memmove(r->ptr, s->ptr, s->bytes);
            r->done = r->bytes + 1; s->done = s->bytes + 1; found = 1;
            break;
        }
        if (!found) { fprintf(stderr, "nccl stub: send to self without a matching receive in the group\n"); g_nops = 0; return 3; }
    }
    for (int i = 0; i < g_nops; ++i)
        if (!g_ops[i].send && g_ops[i].peer == g_ops[i].comm->rank && !g_ops[i].done) { fprintf(stderr, "nccl stub: receive from self without a send\n"); g_nops = 0; return 3; }
    /* remote transfers: each message = 8-byte length header + payload; done == bytes + 1 marks completion */
    for (;;) {
        struct pollfd pf[STUB_MAX_OPS]; int idx[STUB_MAX_OPS]; int n = 0, pending = 0;
        for (int i = 0; i < g_nops; ++i) {
            StubOp* o = &g_ops[i];
            if (o->done == o->bytes + 1) continue;
            ++pending;
            if (!is_head(i)) continue;
            pf[n].fd = o->comm->fd[o->peer]; pf[n].events = o->send ? POLLOUT : POLLIN; pf[n].revents = 0; idx[n++] = i;
        }
        if (!pending) break;
        if (poll(pf, (nfds_t)n, 20000) <= 0) { fprintf(stderr, "nccl stub: rank %d timed out with %d transfers pending\n", g_ops[0].comm->rank, pending); g_nops = 0; return 3; }
        for (int k = 0; k < n; ++k) {
            if (!(pf[k].revents & (POLLIN | POLLOUT | POLLHUP | POLLERR))) continue;
            StubOp* o = &g_ops[idx[k]];
            if (!o->header_done) {
                uint64_t hdr = o->bytes;
                if (o->send) { if (write(pf[k].fd, &hdr, 8) != 8) { g_nops = 0; return 3; } }
                else {
                    if (read(pf[k].fd, &hdr, 8) != 8) { fprintf(stderr, "nccl stub: peer %d closed\n", o->peer); g_nops = 0; return 3; }
                    if (hdr != o->bytes) { fprintf(stderr, "nccl stub: rank %d expects %zu bytes from %d, peer sends %llu\n", o->comm->rank, o->bytes, o->peer, (unsigned long long)hdr); g_nops = 0; return 3; }
                }
                o->header_done = 1;
                continue;
            }
            size_t left = o->bytes - o->done; if (left > (1u << 16)) left = 1u << 16;
            ssize_t m = o->send ? write(pf[k].fd, o->ptr + o->done, left) : read(pf[k].fd, o->ptr + o->done, left);
            if (m < 0 && (errno == EAGAIN || errno == EINTR)) continue;
            if (m <= 0) { fprintf(stderr, "nccl stub: transfer with %d broke\n", o->peer); g_nops = 0; return 3; }
            o->done += (size_t)m;
            if (o->done == o->bytes) o->done = o->bytes + 1;
        }
    }
    g_nops = 0;
    return 0;
}

int ncclGroupEnd(void) {
    ++g_ends;
    if (g_depth <= 0) return 4;
    if (--g_depth > 0) return 0;
    return progress_all();
}
