// rccl_arena.hip -- the path's ONE collective behind the C ABI: broadcast of a packed weight arena from one rank to all others over
// RCCL / xGMI (SURVEY.md 8b "rccl_bcast_arena", 8e).  Host code only.  librccl is opened lazily with dlopen, so libidmvton_hip.so
// itself has no link-time dependency on RCCL and single-GPU users never load it.
//
// The reference has no counterpart (inference.py loads every checkpoint in every process, :232-274): with one process per GPU the
// weights are materialised by rank 0 and reach the other ranks in <= 512 MiB ncclBroadcast pieces -- xGMI is point-to-point, so a
// broadcast is a ring over per-link bandwidth and gains nothing from larger messages, while smaller pieces bound the staging memory.
#include <dlfcn.h>
#include <string.h>
#include "common.cuh"

namespace {
typedef struct { char internal[128]; } nccl_uid;                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_comm_init)(void**, int, nccl_uid, int);
typedef int (*fn_bcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
struct Rccl { void* h; fn_get_uid get_uid; fn_comm_init comm_init; fn_bcast bcast; fn_destroy destroy; fn_errstr errstr; };
Rccl g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

int load_rccl() {
    if (g_rccl.h) return IDMVTON_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return idmvton_set_error(IDMVTON_E_LAUNCH, "rccl: cannot open librccl.so: %s", dlerror());
    Rccl r;
    r.h = h;
    r.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    r.comm_init = (fn_comm_init)dlsym(h, "ncclCommInitRank");
    r.bcast = (fn_bcast)dlsym(h, "ncclBroadcast");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!r.get_uid || !r.comm_init || !r.bcast || !r.destroy) return idmvton_set_error(IDMVTON_E_LAUNCH, "rccl: librccl.so lacks a required symbol");
    g_rccl = r;
    return IDMVTON_OK;
}
int rccl_fail(const char* what, int rc) {
    return idmvton_set_error(IDMVTON_E_LAUNCH, "rccl: %s failed (%d: %s)", what, rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
}
}  // namespace

extern "C" int idmvton_rccl_unique_id(void* id128) {
    CHECK_ARG(id128 != nullptr, IDMVTON_E_ARG, "rccl_unique_id: null buffer");
    if (int e = load_rccl()) return e;
    nccl_uid id;
    if (int rc = g_rccl.get_uid(&id)) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, 128);
    return IDMVTON_OK;
}

extern "C" int idmvton_rccl_comm_init(const void* id128, int rank, int world, void** comm_out) {
    CHECK_ARG(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, IDMVTON_E_ARG, "rccl_comm_init: rank %d of %d", rank, world);
    if (int e = load_rccl()) return e;
    nccl_uid id;
    memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    if (int rc = g_rccl.comm_init(&comm, world, id, rank)) return rccl_fail("ncclCommInitRank", rc);
    *comm_out = comm;
    return IDMVTON_OK;
}

extern "C" int idmvton_rccl_bcast_arena(void* comm, void* buf, uint64_t bytes, int root, uint64_t chunk_bytes, void* stream) {
    CHECK_ARG(comm && buf && bytes > 0 && root >= 0, IDMVTON_E_ARG, "rccl_bcast_arena: comm=%p buf=%p bytes=%llu root=%d", comm, buf, (unsigned long long)bytes, root);
    if (int e = load_rccl()) return e;
    if (chunk_bytes == 0) chunk_bytes = 512ull << 20;
    for (uint64_t off = 0; off < bytes; off += chunk_bytes) {
        const uint64_t n = bytes - off < chunk_bytes ? bytes - off : chunk_bytes;
        char* p = (char*)buf + off;
        if (int rc = g_rccl.bcast(p, p, (size_t)n, /*ncclUint8*/ 1, root, comm, (hipStream_t)stream)) return rccl_fail("ncclBroadcast", rc);
    }
    return IDMVTON_OK;
}

extern "C" int idmvton_rccl_comm_destroy(void* comm) {
    if (!comm) return IDMVTON_OK;
    if (int e = load_rccl()) return e;
    if (int rc = g_rccl.destroy(comm)) return rccl_fail("ncclCommDestroy", rc);
    return IDMVTON_OK;
}
