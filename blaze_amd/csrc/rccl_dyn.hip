#include "rccl_dyn.hpp"

#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace blz {

const RcclApi* rccl_api() {
    static std::mutex mu;
    static RcclApi api;
    static int state = 0;  // 0 untried, 1 ok, -1 failed
    std::lock_guard<std::mutex> lk(mu);
    if (state == 1) return &api;
    if (state == -1) {
        set_last_error("RCCL is not available in this process (librccl.so.1 could not be loaded)");
        return nullptr;
    }
    // The RCCL that sits next to the HIP runtime THIS library is bound to.  A process can carry two ROCm stacks
    // (PyTorch bundles its own libamdhip64 / libhsa-runtime64 / librccl under the system sonames); an RCCL from
    // the other stack talks to an HSA runtime that was never initialised ("no ROCm-capable device is detected").
    void* lib = nullptr;
    std::string dir;
    Dl_info info;
    if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        dir = info.dli_fname;
        size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    std::vector<std::string> names;
    if (!dir.empty()) { names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& name : names) {
        lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (lib) {
            BLZ_LOG(2, "RCCL: %s", name.c_str());
            snprintf(api.path, sizeof(api.path), "%s", name.c_str());
            break;
        }
    }
    if (!lib) {
        state = -1;
        set_last_error("RCCL is not available: %s", dlerror());
        return nullptr;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(lib, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(lib, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(lib, "ncclGetErrorString");
    api.GroupStart = (decltype(api.GroupStart))dlsym(lib, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(lib, "ncclGroupEnd");
    api.CommAbort = (decltype(api.CommAbort))dlsym(lib, "ncclCommAbort");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString || !api.GroupStart ||
        !api.GroupEnd) {
        state = -1;
        set_last_error("RCCL library lacks an expected entry point");
        return nullptr;
    }
    state = 1;
    return &api;
}

}  // namespace blz
