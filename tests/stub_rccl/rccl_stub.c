/*
 * Stand-in for librccl on a machine without GPUs (tests/test_rccl_stub.py): the five symbols ssmtoybox_amd/csrc/ssmq_comm.hip
 * binds, with collective semantics through files in a directory named by the unique id.  Buffers are HOST memory
 * (SSMQ_COMM_HOST_STAGING=1 in libssmq).  TEST INFRASTRUCTURE: reduces in rank order, so every rank gets the same bits.
 *   RCCL_STUB_FAIL_RANK=r   ncclCommInitRank fails on rank r (exercises the all-ranks fallback of mcshard.open_comm)
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <time.h>
#include <sys/stat.h>

#define ID_BYTES 128
typedef struct { char internal[ID_BYTES]; } ncclUniqueId;
typedef struct { int rank, world; long seq; char dir[200]; } stub_comm;
typedef stub_comm *ncclComm_t;

static void nap(void) { struct timespec t = {0, 2000000}; nanosleep(&t, NULL); }
static int exists(const char *p) { struct stat st; return stat(p, &st) == 0; }

int ncclGetUniqueId(ncclUniqueId *id) {
    const char *tmp = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, ID_BYTES, "%s/rccl_stub_%d_%ld", tmp, (int)getpid(), (long)time(NULL));
    return 0;
}

static int wait_all(const stub_comm *c, const char *stem, long seq, double timeout_s) {
    char p[300];
    for (int r = 0; r < c->world; ++r) {
        snprintf(p, sizeof p, "%s/%s.%ld.%d", c->dir, stem, seq, r);
        double waited = 0.0;
        while (!exists(p)) {
            nap();
            waited += 0.002;
            if (waited > timeout_s) return 1;
        }
    }
    return 0;
}

static int put(const stub_comm *c, const char *stem, long seq, const void *data, size_t n) {
    char p[300], t[320];
    snprintf(p, sizeof p, "%s/%s.%ld.%d", c->dir, stem, seq, c->rank);
    snprintf(t, sizeof t, "%s.tmp", p);
    FILE *f = fopen(t, "wb");
    if (!f) return 1;
    if (n && fwrite(data, 1, n, f) != n) { fclose(f); return 1; }
    fclose(f);
    return rename(t, p) != 0;          /* atomic: a reader sees the whole file or none */
}

int ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
    const char *fail = getenv("RCCL_STUB_FAIL_RANK");
    if (fail && atoi(fail) == rank) return 2;          /* ncclSystemError */
    stub_comm *c = (stub_comm *)calloc(1, sizeof(stub_comm));
    c->rank = rank; c->world = world; c->seq = 0;
    snprintf(c->dir, sizeof c->dir, "%s", id.internal);
    mkdir(c->dir, 0700);
    if (put(c, "join", 0, "", 0)) return 2;
    /* collective: returns once every rank has joined (as ncclCommInitRank does); a rank that never arrives blocks the others */
    if (wait_all(c, "join", 0, fail ? 5.0 : 120.0)) { free(c); return 6; }
    *comm = c;
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t c, void *stream) {
    (void)stream;
    if (dtype != 8 || (op != 0 && op != 2)) return 4;   /* ncclFloat64; ncclSum / ncclMax */
    const long seq = ++c->seq;
    if (put(c, "ar", seq, send, count * sizeof(double))) return 2;
    if (wait_all(c, "ar", seq, 120.0)) return 6;
    double *acc = (double *)recv, *tmp = (double *)malloc(count * sizeof(double) + 8);
    char p[300];
    for (int r = 0; r < c->world; ++r) {
        snprintf(p, sizeof p, "%s/ar.%ld.%d", c->dir, seq, r);
        FILE *f = fopen(p, "rb");
        if (!f || fread(tmp, sizeof(double), count, f) != count) { if (f) fclose(f); free(tmp); return 2; }
        fclose(f);
        for (size_t i = 0; i < count; ++i)
            acc[i] = r == 0 ? tmp[i] : (op == 0 ? acc[i] + tmp[i] : (tmp[i] > acc[i] ? tmp[i] : acc[i]));
    }
    free(tmp);
    /* everyone has written seq, i.e. finished reading seq - 1: this rank's file of seq - 1 can go */
    snprintf(p, sizeof p, "%s/ar.%ld.%d", c->dir, seq - 1, c->rank);
    unlink(p);
    return 0;
}

int ncclCommDestroy(ncclComm_t c) {
    char p[300];
    /* the caller's last collective was a barrier: nobody reads older files any more; the last files are left to the rmdir below */
    snprintf(p, sizeof p, "%s/join.0.%d", c->dir, c->rank);
    unlink(p);
    put(c, "bye", 0, "", 0);
    if (c->rank == 0 && !wait_all(c, "bye", 0, 30.0)) {
        char cmd[300];
        snprintf(cmd, sizeof cmd, "rm -rf '%s'", c->dir);
        if (system(cmd)) {}
    }
    free(c);
    return 0;
}

const char *ncclGetErrorString(int rc) { return rc == 0 ? "success" : rc == 6 ? "stub: a rank did not arrive" : "stub error"; }

/* only this stand-in exports the marker: libssmq honours SSMQ_COMM_HOST_STAGING=1 (host pointers handed to ncclAllReduce) only
 * when the library it loaded has it (csrc/ssmq_comm.hip: host_staging) */
int ssmq_stub_rccl_marker(void) { return 1; }
