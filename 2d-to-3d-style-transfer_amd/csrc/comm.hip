// comm.hip -- the one collective of the multi-GPU step, for callers of the C ABI that do not bring torch.distributed:
// SUM all-reduce of the flat [texture grad | vertex grad] fp32 buffer over RCCL (xGMI inside a node), one rank per GPU
// (the reference has no multi-GPU path at all).  The Python host keeps using torch.distributed
// (backend "nccl" IS RCCL on ROCm) -- both end in the same ncclAllReduce.
//
// RCCL is bound at run time (dlopen + dlsym on first use), so libst3d.so carries no link-time dependency on it and a
// process that already holds a copy of RCCL (PyTorch ships one) keeps exactly one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
            r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.lib, "ncclAllReduce"));
            r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
        }
    }
    if (!r.lib || !r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) return nullptr;
    return &r;
}

#define ST3D_NCCL(r, call)                                                                                     \
    do {                                                                                                       \
        ncclResult_t e_ = (call);                                                                              \
        if (e_ != ncclSuccess) {                                                                               \
            st3d::set_error("%s: %s failed: %s", __func__, #call, (r)->error_string ? (r)->error_string(e_) : "?"); \
            return ST3D_E_HIP;                                                                                 \
        }                                                                                                      \
    } while (0)

}  // namespace

struct st3d_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int st3d_comm_unique_id(unsigned char id_out[ST3D_COMM_ID_BYTES]) {
    ST3D_CHECK_ARG(id_out);
    static_assert(ST3D_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    Rccl *r = rccl();
    if (!r) { st3d::set_error("st3d_comm_unique_id: librccl.so not found"); return ST3D_E_STATE; }
    ncclUniqueId id;
    ST3D_NCCL(r, r->get_unique_id(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ST3D_OK;
}

extern "C" int st3d_comm_init(st3d_comm **out, int rank, int world, const unsigned char unique_id[ST3D_COMM_ID_BYTES]) {
    ST3D_CHECK_ARG(out && unique_id && world >= 1 && rank >= 0 && rank < world);
    Rccl *r = rccl();
    if (!r) { st3d::set_error("st3d_comm_init: librccl.so not found"); return ST3D_E_STATE; }
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    st3d_comm *c = new st3d_comm{nullptr, rank, world};
    ncclResult_t e = r->comm_init_rank(&c->comm, world, id, rank);       // uses the calling thread's current HIP device
    if (e != ncclSuccess) {
        st3d::set_error("st3d_comm_init: ncclCommInitRank failed: %s", r->error_string ? r->error_string(e) : "?");
        delete c;
        return ST3D_E_HIP;
    }
    *out = c;
    return ST3D_OK;
}

extern "C" int st3d_allreduce_sum_f32(st3d_comm *comm, float *buf, size_t n, st3d_stream_t stream) {
    ST3D_CHECK_ARG(comm && buf && n > 0);
    Rccl *r = rccl();
    if (!r) { st3d::set_error("st3d_allreduce_sum_f32: librccl.so not found"); return ST3D_E_STATE; }
    ST3D_NCCL(r, r->all_reduce(buf, buf, n, ncclFloat32, ncclSum, comm->comm, st3d::as_stream(stream)));
    return ST3D_OK;
}

extern "C" int st3d_comm_destroy(st3d_comm *comm) {
    if (!comm) return ST3D_OK;
    Rccl *r = rccl();
    if (r && comm->comm) (void)r->comm_destroy(comm->comm);
    delete comm;
    return ST3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// roctx ranges (the reference has no tracing; its tqdm bars become named ranges a profiler can show).
// The roctx library is bound at run time like RCCL, and only when ST3D_ROCTX=1: without it both calls return at once.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
};
Roctx &roctx() {
    static Roctx r;
    if (!r.tried) {
        r.tried = true;
        const char *e = getenv("ST3D_ROCTX");
        if (e && e[0] == '1') {
            // (rocprofv3 listens to the rocprofiler-sdk flavour of roctx; libroctx64 is the roctracer one)
            for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                     "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"}) {
                if (void *lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                    r.push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
                    r.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
                    if (r.push && r.pop) break;
                    r.push = nullptr; r.pop = nullptr;
                }
            }
        }
    }
    return r;
}
}  // namespace

extern "C" int st3d_trace_push(const char *name) {
    Roctx &r = roctx();
    if (r.push && name) r.push(name);
    return ST3D_OK;
}

extern "C" int st3d_trace_pop(void) {
    Roctx &r = roctx();
    if (r.pop) r.pop();
    return ST3D_OK;
}

extern "C" int st3d_trace_enabled(void) { return roctx().push ? 1 : 0; }
