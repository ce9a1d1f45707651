// rccl_dyn.hpp — RCCL, bound at run time.  The read-back exchange of a multi-GPU group (capi.hip group_gather) is one grouped
// RCCL operation over xGMI when the collective library is there; the library is NOT a link-time dependency of
// libchunky_hip.so: a JVM (or a 1-GPU box) without librccl still loads this library and renders, and a process that already
// holds an RCCL (PyTorch ships its own, same soname) gets that one instead of a second copy.  Nothing of RCCL is needed to
// BUILD this library either: the handful of types and enumerators the calls take are declared here (their values are ABI:
// rccl.h / nccl.h have not changed them), every function is looked up with dlsym.  Where <rccl/rccl.h> is installed, defining
// CHUNKY_CHECK_RCCL_HEADER makes the compiler compare the two (tests/test_cabi.py does).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#if defined(CHUNKY_CHECK_RCCL_HEADER)
#include <rccl/rccl.h>
static_assert(ncclSuccess == 0 && ncclSystemError == 2 && ncclRemoteError == 6 && ncclFloat == 7 && ncclSum == 0, "RCCL's enumerators moved");
static_assert(sizeof(ncclComm_t) == sizeof(void*) && sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int),
              "RCCL's types changed size");
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include <cstdlib>
#include <mutex>
#include <string>

namespace chunky {

struct RcclApi {
    void* handle = nullptr;
    std::string where;  // what dlopen was given
    std::string error;  // why the library is unusable (empty when it is usable)
    int version = 0;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    bool usable() const { return handle != nullptr && error.empty(); }
    const char* str(ncclResult_t r) const { return GetErrorString ? GetErrorString(r) : "?"; }
};

// The process-wide binding: tried once.  CHUNKY_RCCL_LIB names the file to load (a deployment with its own ROCm; also how the
// tests make the load fail); otherwise the soname — which finds an RCCL the process already holds — then ROCm's usual place.
inline const RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("CHUNKY_RCCL_LIB");
        const char* names[] = {env && *env ? env : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const int first = env && *env ? 0 : 1, last = env && *env ? 1 : 4;  // an explicit file is the only candidate
        std::string tried;
        for (int i = first; i < last && !api.handle; i++) {
            api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
            if (api.handle) {
                api.where = names[i];
            } else {
                const char* why = dlerror();
                tried += std::string(tried.empty() ? "" : "; ") + (why ? why : names[i]);
            }
        }
        if (!api.handle) {
            api.error = "RCCL not loaded: " + tried;
            return;
        }
        bool all = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(api.handle, name);
            if (!p) {
                all = false;
                api.error += std::string(api.error.empty() ? "RCCL lacks " : ", ") + name;
            }
            return p;
        };
        api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.CommGetAsyncError = (decltype(api.CommGetAsyncError))sym("ncclCommGetAsyncError");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
        if (all && api.GetVersion(&api.version) != ncclSuccess) api.version = 0;
    });
    return api;
}

}  // namespace chunky
