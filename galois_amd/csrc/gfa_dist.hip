// gfa_dist.hip -- the distributed four-step transform behind the C-ABI, collective included.
//
// SURVEY.md section 8(b) asks for `gf_ntt_dist(comm, ...)` with the RCCL all-to-all INSIDE the library, so that a host in any
// language can run BASELINE config C5 (one 2^26-point Goldilocks transform over the GPUs of a node): one process per GPU,
// the caller creates the communicator (ncclCommInitRank) and passes it in.  The Python front end keeps using
// torch.distributed for the same exchange (galois_amd/_dist.py) because torch owns its communicator.
//
// RCCL is bound at first use with dlsym: first in the symbols already loaded into the process (a PyTorch-ROCm process has
// its own librccl), then from librccl.so -- libgalois_amd.so itself has no link-time dependency on it.
#include <dlfcn.h>

#include "gfa_internal.h"

using namespace gfa;

namespace {

typedef int (*nccl_alltoall_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_sendrecv_fn)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_recv_fn)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_fn)(void);
typedef const char *(*nccl_err_fn)(int);

struct Rccl {
    nccl_alltoall_fn alltoall = nullptr;
    nccl_sendrecv_fn send = nullptr;
    nccl_recv_fn recv = nullptr;
    nccl_group_fn group_start = nullptr, group_end = nullptr;
    nccl_err_fn errstr = nullptr;
    bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int bind_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        void *h = RTLD_DEFAULT; // (a null handle: the symbols already visible in the process)
        bool found = dlsym(h, "ncclSend") != nullptr;
        if (!found) {
            for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) { found = true; break; }
            }
        }
        if (found) {
            g_rccl.alltoall = (nccl_alltoall_fn)dlsym(h, "ncclAllToAll");
            g_rccl.send = (nccl_sendrecv_fn)dlsym(h, "ncclSend");
            g_rccl.recv = (nccl_recv_fn)dlsym(h, "ncclRecv");
            g_rccl.group_start = (nccl_group_fn)dlsym(h, "ncclGroupStart");
            g_rccl.group_end = (nccl_group_fn)dlsym(h, "ncclGroupEnd");
            g_rccl.errstr = (nccl_err_fn)dlsym(h, "ncclGetErrorString");
        }
    }
    if (!g_rccl.alltoall && !(g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end)) {
        set_error("gfa_ntt_dist: RCCL (librccl.so) is not available in this process");
        return GFA_ERR_UNSUPPORTED;
    }
    return GFA_OK;
}

int nccl_fail(int rc, const char *what)
{
    set_error(std::string(what) + " failed: " + (g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error " + std::to_string(rc)));
    return GFA_ERR_HIP;
}

// every rank sends block s (count elements) of `send` to rank s and receives block s of `recv` from rank s
int all_to_all(const void *send, void *recv, size_t count, int dtype, void *comm, int world, hipStream_t st)
{
    const int nccl_type = dtype == GFA_U32 ? 3 /* ncclUint32 */ : 5 /* ncclUint64 */;
    const size_t esz = dtype == GFA_U32 ? 4 : 8;
    int rc;
    if (g_rccl.alltoall) {
        if ((rc = g_rccl.alltoall(send, recv, count, nccl_type, comm, st))) return nccl_fail(rc, "ncclAllToAll");
        return GFA_OK;
    }
    if ((rc = g_rccl.group_start())) return nccl_fail(rc, "ncclGroupStart");
    for (int s = 0; s < world; s++) {
        if ((rc = g_rccl.send((const char *)send + (size_t)s * count * esz, count, nccl_type, s, comm, st))) return nccl_fail(rc, "ncclSend");
        if ((rc = g_rccl.recv((char *)recv + (size_t)s * count * esz, count, nccl_type, s, comm, st))) return nccl_fail(rc, "ncclRecv");
    }
    if ((rc = g_rccl.group_end())) return nccl_fail(rc, "ncclGroupEnd");
    return GFA_OK;
}

// The overlapped exchange's side resources, one set per device, created at first use and kept for the process: a non-blocking
// side stream, the three events that order it against the caller's stream, and a lock held while one call ENQUEUES its exchange
// (host-side work only) so that two host threads cannot interleave their records / waits on the shared events.
struct SideSet {
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    std::mutex issue;
};
int side_set(SideSet **out)
{
    static std::mutex mu;
    static SideSet sets[64];
    int dev = 0;
    GFA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) { set_error("gfa_ntt_dist: device ordinal out of range"); return GFA_ERR_UNSUPPORTED; }
    SideSet &ss = sets[dev];
    if (!ss.stream) GFA_HIP(hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking));
    for (auto &e : ss.ev)
        if (!e) GFA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = &ss;
    return GFA_OK;
}

int check_dist_args(gfa_field_t *f, const void *a, const void *b, void *comm, int rank, int world, int64_t n1, int64_t n2, int dtype,
                    const char *what)
{
    if (!f || !a || !b || !comm || world < 1 || rank < 0 || rank >= world || n1 < 2 || n2 < 2 || (n1 & (n1 - 1)) || (n2 & (n2 - 1)) ||
        n1 % world || n2 % world || a == b) {
        set_error(std::string(what) + ": bad arguments (power-of-two n1, n2 divisible by the number of ranks; distinct buffers)");
        return GFA_ERR_INVALID;
    }
    const FieldDev &c = f->calc;
    if (dtype != (c.kind == KIND_PRIME32 ? GFA_U32 : GFA_U64) || !(c.kind == KIND_PRIME32 || c.kind == KIND_PRIME64 || c.kind == KIND_GOLDILOCKS)) {
        set_error(std::string(what) + ": prime fields, native device width (uint32 for p < 2^32, else uint64)");
        return GFA_ERR_UNSUPPORTED;
    }
    return GFA_OK;
}

} // namespace

extern "C" {

int gfa_ntt_dist(gfa_field_t *f, void *nccl_comm, int rank, int world, const void *local_cols, void *out_rows, int64_t n1, int64_t n2,
                 uint64_t omega, int dtype, gfa_stream_t stream)
{
    int rc;
    if ((rc = check_dist_args(f, local_cols, out_rows, nccl_comm, rank, world, n1, n2, dtype, "gfa_ntt_dist"))) return rc;
    if ((rc = bind_rccl())) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t cols = n2 / world, rows = n1 / world, n_total = n1 * n2;
    const size_t esz = dtype == GFA_U32 ? 4 : 8;
    const size_t bytes = (size_t)(n1 * cols) * esz;
    // The column pass and the exchange cost about the same (C5, 8 ranks: 0.145 ms of kernels, ~0.11 ms on the links), so the
    // columns are transformed in two sub-blocks and the first one travels while the second is computed: still ONE logical
    // all-to-all, issued as two grouped send / recv halves on a side stream.  GFA_DIST_NSUB=1: one ncclAllToAll, no overlap.
    static const int nsub_env = [] { const char *e = getenv("GFA_DIST_NSUB"); return e ? atoi(e) : 2; }();
    const bool p2p = g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end;
    const int nsub = (p2p && nsub_env >= 2 && cols % 2 == 0 && cols >= 64) ? 2 : 1;
    const int64_t csub = cols / nsub;
    SideSet *ss = nullptr;
    if (nsub > 1 && (rc = side_set(&ss))) return rc;
    char *work = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&work, 2 * bytes, st));
    char *a = work, *recv = work + bytes;
    u64 omega_n2 = 0;
    HostArith::pow(f->calc, omega, n1, &omega_n2);
    // (1) columns + twiddle, sub-block s written as its own compact (n1 x csub) array, (2) the exchange: rows [r*rows, (r+1)*rows)
    // of sub-block s go to rank r and land at recv[peer][s][k1_local][c], (3) rows, read from those chunks in place
    rc = GFA_OK;
    if (nsub == 1) {
        rc = gfa_ntt_columns(f, local_cols, a, n1, cols, (int64_t)rank * cols, n_total, omega, dtype, stream);
        if (!rc) rc = all_to_all(a, recv, (size_t)(rows * cols), dtype, nccl_comm, world, st);
    } else {
        const int nccl_type = dtype == GFA_U32 ? 3 /* ncclUint32 */ : 5 /* ncclUint64 */;
        const size_t blk = (size_t)(rows * csub); // elements per (peer, sub-block) chunk
        std::lock_guard<std::mutex> issue(ss->issue);
        hipStream_t side = ss->stream;
        bool side_used = false;
        for (int s = 0; s < nsub && !rc; s++) {
            char *as = a + (size_t)s * (size_t)(n1 * csub) * esz;
            rc = gfa_ntt_columns_pitched(f, (const char *)local_cols + (size_t)s * csub * esz, cols, as, csub, n1, csub,
                                         (int64_t)rank * cols + s * csub, n_total, omega, dtype, stream);
            if (rc) break;
            hipError_t he = hipEventRecord(ss->ev[s], st);
            if (he == hipSuccess) he = hipStreamWaitEvent(side, ss->ev[s], 0);
            if (he != hipSuccess) { rc = hip_fail(he, "gfa_ntt_dist: event"); break; }
            int nr;
            if ((nr = g_rccl.group_start())) { rc = nccl_fail(nr, "ncclGroupStart"); break; }
            side_used = true;
            for (int p = 0; p < world && !rc; p++) {
                if ((nr = g_rccl.send(as + (size_t)p * blk * esz, blk, nccl_type, p, nccl_comm, side))) rc = nccl_fail(nr, "ncclSend");
                else if ((nr = g_rccl.recv(recv + ((size_t)p * nsub + s) * blk * esz, blk, nccl_type, p, nccl_comm, side))) rc = nccl_fail(nr, "ncclRecv");
            }
            if ((nr = g_rccl.group_end()) && !rc) rc = nccl_fail(nr, "ncclGroupEnd");
        }
        if (side_used) {
            // also on a failed step: whatever already runs on the side stream may still touch `work`, which goes back to the
            // pool ordered on `st` alone -- `st` joins the side stream before anything else (the free included) is enqueued
            hipError_t he = hipEventRecord(ss->ev[2], side);
            if (he == hipSuccess) he = hipStreamWaitEvent(st, ss->ev[2], 0);
            if (he != hipSuccess) {
                (void)hipStreamSynchronize(side);
                if (!rc) rc = hip_fail(he, "gfa_ntt_dist: event");
            }
        }
    }
    if (!rc) {
        rc = gfa_ntt_chunked(f, recv, out_rows, n2, rows, omega_n2, 0, csub, rows * csub, csub, 0, 0, 0, dtype, stream);
        if (rc == GFA_ERR_UNSUPPORTED) {
            // chunk sizes the row kernel does not take in place (chunks below its granule, n2 > 2^20): re-lay the chunks
            // recv[peer][s][k1_local][c] out as whole rows in the column buffer (free since the exchange) -- as galois_amd/_dist.py does
            rc = GFA_OK;
            for (int ch = 0; ch < world * nsub && !rc; ch++) {
                const hipError_t ce = hipMemcpy2DAsync(a + (size_t)ch * csub * esz, (size_t)n2 * esz, recv + (size_t)ch * rows * csub * esz,
                                                       (size_t)csub * esz, (size_t)csub * esz, (size_t)rows, hipMemcpyDeviceToDevice, st);
                if (ce != hipSuccess) rc = hip_fail(ce, "hipMemcpy2DAsync (row re-layout)");
            }
            if (!rc) rc = gfa_ntt(f, a, out_rows, n2, rows, omega_n2, 0, dtype, stream);
        }
    }
    const hipError_t fe = gfa::scratch_free(work, st);
    if (rc) return rc;
    GFA_HIP(fe);
    return GFA_OK;
}

int gfa_intt_dist(gfa_field_t *f, void *nccl_comm, int rank, int world, const void *local_rows, void *out_cols, int64_t n1, int64_t n2,
                  uint64_t omega, int scale_by_n_inverse, int dtype, gfa_stream_t stream)
{
    int rc;
    if ((rc = check_dist_args(f, local_rows, out_cols, nccl_comm, rank, world, n1, n2, dtype, "gfa_intt_dist"))) return rc;
    if ((rc = bind_rccl())) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t cols = n2 / world, rows = n1 / world, n_total = n1 * n2;
    const size_t bytes = (size_t)(n1 * cols) * (dtype == GFA_U32 ? 4 : 8);
    u64 omega_inv = 0, w_rows = 0;
    if (!HostArith::inv(f->calc, omega, &omega_inv)) { set_error("gfa_intt_dist: omega is not invertible"); return GFA_ERR_INVALID; }
    HostArith::pow(f->calc, omega_inv, n1, &w_rows);
    char *work = nullptr;
    GFA_HIP(gfa::scratch_alloc((void **)&work, 2 * bytes, st));
    void *send = work, *recv = work + bytes;
    // (1) rows, written straight into the send buffer send[s][k1_local][c]; (2) the one exchange; (3) pre-twiddle + columns + 1/N
    rc = gfa_ntt_chunked(f, local_rows, send, n2, rows, w_rows, 0, 0, 0, 0, cols, rows * cols, cols, dtype, stream);
    if (rc == GFA_ERR_UNSUPPORTED) {
        // whole rows into `recv` (not yet in use), then cut into the per-peer chunks send[s][k1_local][c]
        const size_t esz = dtype == GFA_U32 ? 4 : 8;
        rc = gfa_ntt(f, local_rows, recv, n2, rows, w_rows, 0, dtype, stream);
        for (int s = 0; s < world && !rc; s++) {
            const hipError_t ce = hipMemcpy2DAsync((char *)send + (size_t)s * rows * cols * esz, (size_t)cols * esz,
                                                   (const char *)recv + (size_t)s * cols * esz, (size_t)n2 * esz, (size_t)cols * esz,
                                                   (size_t)rows, hipMemcpyDeviceToDevice, st);
            if (ce != hipSuccess) rc = hip_fail(ce, "hipMemcpy2DAsync (chunk re-layout)");
        }
    }
    if (!rc) rc = all_to_all(send, recv, (size_t)(rows * cols), dtype, nccl_comm, world, st);
    if (!rc) rc = gfa_ntt_columns_inv(f, recv, out_cols, n1, cols, (int64_t)rank * cols, n_total, omega_inv, scale_by_n_inverse, dtype, stream);
    const hipError_t fe = gfa::scratch_free(work, st);
    if (rc) return rc;
    GFA_HIP(fe);
    return GFA_OK;
}

} // extern "C"
