/* rccl_stub.c — a stand-in for librccl, TEST INFRASTRUCTURE ONLY (tests/test_gpu_rccl_transport.py): it lets a 1-GPU box watch
 * libchunky_hip's read-back exchange survive an RCCL that breaks after the communicator was created.  libchunky_hip binds RCCL
 * with dlopen (csrc/rccl_dyn.hpp; CHUNKY_RCCL_LIB names the file), so the product needs no hook for this.  The stub "creates"
 * communicators for any device list and then misbehaves as RCCL_STUB_MODE says:
 *   fail_send   the first ncclSend / ncclReduce returns ncclSystemError
 *   fail_end    ncclGroupEnd returns ncclSystemError
 *   async       every call reports success and moves nothing; ncclCommGetAsyncError then reports ncclRemoteError
 * Built with -DRCCL_STUB_OMIT_REDUCE it lacks ncclReduce altogether (an RCCL too old for the library).  Plain ints and pointers:
 * the enum values are those of <rccl/rccl.h> (ncclSuccess 0, ncclSystemError 2, ncclRemoteError 6). */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef void* comm_t;
static int calls_send, calls_end;
static int mode_is(const char* m) {
    const char* e = getenv("RCCL_STUB_MODE");
    return e && strcmp(e, m) == 0;
}
int ncclGetVersion(int* v) { *v = 29999; return 0; }
int ncclCommInitAll(comm_t* comms, int n, const int* devs) {
    (void)devs;
    for (int i = 0; i < n; i++) comms[i] = (comm_t)(size_t)(0x1000 + i);
    return 0;
}
int ncclCommDestroy(comm_t c) { (void)c; return 0; }
int ncclCommAbort(comm_t c) { (void)c; return 0; }
int ncclCommGetAsyncError(comm_t c, int* e) { (void)c; *e = mode_is("async") ? 6 : 0; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 2 ? "unhandled system error (stub)" : (r == 6 ? "remote process exited or there was a network error (stub)" : "stub error")); }
int ncclGroupStart(void) { return 0; }
int ncclGroupEnd(void) { calls_end++; return mode_is("fail_end") ? 2 : 0; }
int ncclSend(const void* b, size_t n, int t, int peer, comm_t c, void* s) { (void)b; (void)n; (void)t; (void)peer; (void)c; (void)s; calls_send++; return mode_is("fail_send") ? 2 : 0; }
int ncclRecv(void* b, size_t n, int t, int peer, comm_t c, void* s) { (void)b; (void)n; (void)t; (void)peer; (void)c; (void)s; return 0; }
#ifndef RCCL_STUB_OMIT_REDUCE
int ncclReduce(const void* a, void* b, size_t n, int t, int op, int root, comm_t c, void* s) { (void)a; (void)b; (void)n; (void)t; (void)op; (void)root; (void)c; (void)s; calls_send++; return mode_is("fail_send") ? 2 : 0; }
#endif
