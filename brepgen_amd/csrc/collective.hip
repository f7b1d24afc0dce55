// The path's ONE collective behind the C ABI (bg_allgather): the all-gather of the finished latents before the
// CPU-side B-rep reconstruction (brepgen_amd/sampling.py: gather_latents -- one flat byte buffer per rank, every rank padded to the same
// size).  The Python host reaches RCCL through torch.distributed (backend "nccl" IS RCCL on ROCm); a host in another language has no
// torch: this entry is a thin ncclAllGather over a communicator the CALLER created (ncclCommInitRank), on the caller's stream, so that
// it has a C-ABI route to the collective without this library owning any process-group state.  RCCL is resolved at the first call
// (dlopen: the library itself does not link against librccl, and a single-GPU deployment never loads it).
#include "bg_common.h"
#include <dlfcn.h>

namespace bg {
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);      // ncclResult_t ncclAllGather(..., ncclDataType_t, ncclComm_t, hipStream_t)
typedef const char* (*nccl_errstr_fn)(int);
nccl_allgather_fn g_allgather = nullptr;
nccl_errstr_fn g_errstr = nullptr;

bool resolve_rccl() {
    if (g_allgather) return true;
    // an already loaded RCCL first (the one the host initialised its communicator with), then the system's
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(h, "ncclGetErrorString"));
    g_allgather = reinterpret_cast<nccl_allgather_fn>(dlsym(h, "ncclAllGather"));
    return g_allgather != nullptr;
}
}  // namespace
}  // namespace bg

extern "C" int bg_allgather(const void* send, void* recv, size_t bytes_per_rank, void* nccl_comm, bg_stream_t stream) {
    using namespace bg;
    BG_REQUIRE(nccl_comm != nullptr, BG_E_ARG, "bg_allgather: null communicator (create one with ncclCommInitRank)");
    BG_REQUIRE(bytes_per_rank == 0 || (send && recv), BG_E_ARG, "bg_allgather: null buffer");
    if (bytes_per_rank == 0) return 0;
    BG_REQUIRE(resolve_rccl(), BG_E_ARG, "bg_allgather: librccl.so not found (%s)", dlerror() ? dlerror() : "dlopen failed");
    const int rc = g_allgather(send, recv, bytes_per_rank, /* ncclUint8 */ 1, nccl_comm, (hipStream_t)stream);
    BG_REQUIRE(rc == 0, 1000 + rc, "bg_allgather: ncclAllGather failed: %s", g_errstr ? g_errstr(rc) : "?");
    return 0;
}
