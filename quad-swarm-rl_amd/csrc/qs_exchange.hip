// qs_exchange.hip - observation exchange between environment shards by peer stores (include/quadswarm_exchange.h).
//
// One process per GPU.  Rank r owns a DATA window [2 slots][world][rows*cols] of the wire type and a FLAG window
// {arrive[2][MAX], ack[MAX]}; both are exported (hipIpcGetMemHandle) and mapped by every peer.  After control step `seq`
//   push    (producer, own stream)   : waits until ack[d] >= seq - 2 for every destination d, then each workgroup copies its
//                                      share of the rank's float32 rows into slot [seq & 1][rank] of destination d's window
//                                      (16-byte stores, a wave writes 1 KB contiguous; float32 -> bfloat16 RNE on the way when
//                                      the wire type says so); the last workgroup per destination raises arrive[seq & 1][rank]
//                                      = seq in d's flag window behind a system-scope release;
//   wait    (consumer)               : one wave polls arrive[seq & 1][*] of its OWN flag window; the kernel boundary behind it
//                                      is the acquire for the readers of the slot;
//   release (consumer)               : stores ack[rank] = seq into every peer's flag window.
// xGMI is point to point: destination d's rows travel over the one link to d, the `world` destinations of one push run
// concurrently (grid.y = world), nothing is forwarded.  gfx950 only; no RCCL in this file.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <unistd.h>

#include "qs_xchg_dev.h"

using namespace qsx;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &m) { g_err = m; return code; }
#define XTRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

// copy / convert this workgroup's share (part of parts) of src[0, n) into dst (wire type).  Vector path: groups of 8 floats
// (two 16-byte loads, one or two 16-byte stores per lane); pointers that are not 16-byte aligned (a row count that is not a
// multiple of 8 puts rank r's rows at an odd offset) take the element-wise path.
__device__ __forceinline__ void copy_range(const float *__restrict__ src, char *__restrict__ dst, long long n, int wire, int part,
    int parts) {
    const bool vec = ((((size_t)src) | ((size_t)dst)) & 15) == 0;
    if (!vec) {
        const long long per = (n + parts - 1) / parts, k0 = per * part, k1 = (k0 + per < n) ? k0 + per : n;
        if (wire == QS_WIRE_BF16) { unsigned short *d2 = (unsigned short *)dst;
            for (long long k = k0 + threadIdx.x; k < k1; k += blockDim.x) st2_wt(d2 + k, f32_to_bf16_rne(src[k])); }
        else { float *d1 = (float *)dst;
            for (long long k = k0 + threadIdx.x; k < k1; k += blockDim.x) st4_wt(d1 + k, __float_as_uint(src[k])); }
        return;
    }
    const long long nvec = n >> 3;                                   // groups of 8 floats
    const long long per = (nvec + parts - 1) / parts, v0 = per * part, v1 = (v0 + per < nvec) ? v0 + per : nvec;
    const float4 *s4 = (const float4 *)src;
    if (wire == QS_WIRE_BF16) {
        uint4 *d4 = (uint4 *)dst;
        for (long long v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
            const float4 a = s4[2 * v], b = s4[2 * v + 1];
            u32x4_t o;
            o.x = f32_to_bf16_rne(a.x) | (f32_to_bf16_rne(a.y) << 16); o.y = f32_to_bf16_rne(a.z) | (f32_to_bf16_rne(a.w) << 16);
            o.z = f32_to_bf16_rne(b.x) | (f32_to_bf16_rne(b.y) << 16); o.w = f32_to_bf16_rne(b.z) | (f32_to_bf16_rne(b.w) << 16);
            st16_wt(d4 + v, o);
        }
        if (part == 0) {
            unsigned short *d2 = (unsigned short *)dst;
            for (long long k = (nvec << 3) + threadIdx.x; k < n; k += blockDim.x) st2_wt(d2 + k, f32_to_bf16_rne(src[k]));
        }
    } else {
        float4 *d4 = (float4 *)dst;
        for (long long v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
            const float4 a = s4[2 * v], b = s4[2 * v + 1];
            st16_wt(d4 + 2 * v, u32x4_t{__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)});
            st16_wt(d4 + 2 * v + 1, u32x4_t{__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)});
        }
        if (part == 0) {
            float *d1 = (float *)dst;
            for (long long k = (nvec << 3) + threadIdx.x; k < n; k += blockDim.x) st4_wt(d1 + k, __float_as_uint(src[k]));
        }
    }
}

// QS_WIRE_Q8: this workgroup's share of `rows` float32 rows [rows][q.D] -> wire rows (q.row_words 4-byte words each), 16 bytes per store
// where source share and destination are 16-byte aligned (always, for the row counts of whole wavefront blocks)
__device__ __forceinline__ void copy_rows_q8(const float *__restrict__ src, char *__restrict__ dst, long long rows, const Q8Dev &q,
    int part, int parts) {
    const long long words = rows * q.row_words;
    if ((((size_t)dst) & 15) == 0 && (words & 3) == 0) {
        const long long nvec = words >> 2, per = (nvec + parts - 1) / parts, v0 = per * part, v1 = (v0 + per < nvec) ? v0 + per : nvec;
        for (long long v = v0 + threadIdx.x; v < v1; v += blockDim.x) {
            unsigned int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long w = 4 * v + u, r = w / q.row_words;
                o[u] = q8_row_word(src + r * q.D, q, (int)(w - r * q.row_words));
            }
            st16_wt(dst + 16 * v, u32x4_t{o[0], o[1], o[2], o[3]});
        }
    } else {
        const long long per = (words + parts - 1) / parts, w0 = per * part, w1 = (w0 + per < words) ? w0 + per : words;
        for (long long w = w0 + threadIdx.x; w < w1; w += blockDim.x) {
            const long long r = w / q.row_words;
            st4_wt(dst + 4 * w, q8_row_word(src + r * q.D, q, (int)(w - r * q.row_words)));
        }
    }
}

__global__ void __launch_bounds__(256) qs_xchg_push_kernel(PushArgs a) {
    const int d = blockIdx.y;                                         // destination rank
    const unsigned long long seq = a.loc->push_seq + 1;               // advanced by the very last workgroup of this launch only
    const int slot = (int)(seq & 1);
    if (threadIdx.x == 0) {
        // flow control: destination d must have released what this slot held before (seq - 2)
        const bool ok = seq <= 2 || poll_ge(&a.mine->ack[d], seq - 2, a.timeout_ticks);   // (d == rank: this rank's own consumer)
        if (!ok) atomicOr(&a.loc->status, (unsigned int)QS_XCHG_ERR_ACK_TIMEOUT);
    }
    __syncthreads();
    const float *src = a.src ? a.src : a.staging[slot];
    if (a.wire == QS_WIRE_Q8) {
        char *dst = a.data_win[d] + (size_t)slot * a.slot_bytes + (size_t)a.rank * (size_t)a.rows * (size_t)a.q8.row_words * 4;
        copy_rows_q8(src, dst, a.rows, a.q8, blockIdx.x, gridDim.x);
    } else {
        const size_t wsz = a.wire == QS_WIRE_BF16 ? 2 : 4;
        char *dst = a.data_win[d] + (size_t)slot * a.slot_bytes + (size_t)a.rank * a.n * wsz;
        copy_range(src, dst, a.n, a.wire, blockIdx.x, gridDim.x);
    }
    wt_drain();                                                       // this thread's (write-through) rows have reached their window ...
    __syncthreads();
    if (threadIdx.x == 0) {
        fence_release_sys(a.fenced);                                  // fenced protocol: every workgroup, before its ticket (qs_xchg_dev.h)
        const unsigned int t = atomicAdd(&a.loc->ticket[d], 1u);
        if (t == gridDim.x - 1) {                                     // ... and this is the last workgroup of destination d
            a.loc->ticket[d] = 0;
            st_sys(&a.flag_win[d]->arrive[slot][a.rank], seq);
            const unsigned int g = atomicAdd(&a.loc->ticket_all, 1u);
            if (g == gridDim.y - 1) { a.loc->ticket_all = 0; __threadfence(); a.loc->push_seq = seq; }
        }
    }
}

struct ReleaseArgs { FlagWin *flag_win[QS_XCHG_MAX_RANKS]; Local *loc; int world, rank, fenced; };
// consumer: wait for the rows of sequence number wait_seq + 1 from every source; with `release` also hand the slot back at once
// (a consumer that does not read the rows in place, e.g. the benchmark, or one that copies them out in this same kernel's shadow)
__global__ void __launch_bounds__(64) qs_xchg_wait_kernel(FlagWin *mine, ReleaseArgs a, unsigned long long timeout_ticks, int release) {
    const unsigned long long seq = a.loc->wait_seq + 1;
    const int slot = (int)(seq & 1), r = threadIdx.x;
    bool ok = true;
    if (r < a.world) ok = poll_ge(&mine->arrive[slot][r], seq, timeout_ticks);
    // (the readers of the rows are later launches on this stream: the invalidate is in place before they start)
    fence_acquire_sys(a.fenced);
    if (__any(!ok) && r == 0) atomicOr(&a.loc->status, (unsigned int)QS_XCHG_ERR_ARRIVE_TIMEOUT);
    if (release && r < a.world) st_sys(&a.flag_win[r]->ack[a.rank], seq);
    if (r == 0) { a.loc->wait_seq = seq; if (release) a.loc->release_seq = seq; }
}

__global__ void __launch_bounds__(64) qs_xchg_release_kernel(ReleaseArgs a) {
    const unsigned long long seq = a.loc->wait_seq;
    const int r = threadIdx.x;
    if (r < a.world) st_sys(&a.flag_win[r]->ack[a.rank], seq);
    if (r == 0) a.loc->release_seq = seq;
}

__global__ void __launch_bounds__(256) qs_obs_pack_kernel(const float *src, char *dst, long long n,
    int wire) { copy_range(src, dst, n, wire, blockIdx.x, gridDim.x); wt_drain(); }
__global__ void __launch_bounds__(256) qs_obs_pack_q8_kernel(const float *src, char *dst, long long rows,
    Q8Dev q) { copy_rows_q8(src, dst, rows, q, blockIdx.x, gridDim.x); wt_drain(); }
// wire rows -> float32 rows (one thread per element)
__global__ void __launch_bounds__(256) qs_obs_unpack_kernel(const char *src, float *dst, long long rows, int cols, int wire, Q8Dev q) {
    const long long n = rows * cols;
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        float v;
        if (wire == QS_WIRE_F32) v = ((const float *)src)[k];
        else if (wire == QS_WIRE_BF16) v = __uint_as_float((unsigned int)((const unsigned short *)src)[k] << 16);
        else {
            const long long r = k / cols;
            const int c = (int)(k - r * cols);
            const char *row = src + r * (long long)q.row_words * 4;
            if (c >= q.q0 && c < q.q1) {
                const int mm = c - q.q0;
                v = (float)((const signed char *)(row + 4 * q.w16))[mm] / q8_scale(q, mm % 6);
            } else {
                const int b = c < q.q0 ? c : c - (q.q1 - q.q0);
                v = __uint_as_float((unsigned int)((const unsigned short *)row)[b] << 16);
            }
        }
        dst[k] = v;
    }
}

struct Blob { hipIpcMemHandle_t data, flags; int32_t pid, device; int32_t pad[2]; };
static_assert(sizeof(hipIpcMemHandle_t) == QS_XCHG_HANDLE_BYTES, "handle size");
static_assert(sizeof(Blob) == QS_XCHG_EXPORT_BYTES, "blob size");

// workgroups per destination: enough 16-byte stores in flight per link, few enough to leave the CUs to the step kernel
int grid_parts(long long n) {
    const long long nvec = n >> 3;
    long long parts = (nvec + 2047) / 2048;   // >= 2048 vectors (32 KB of bf16) per workgroup
    if (parts < 1) parts = 1;
    if (parts > 16) parts = 16;
    return (int)parts;
}

}   // namespace

struct qs_xchg {
    int device = 0, world = 1, rank = 0, wire = QS_WIRE_F32;
    long long n = 0, rows = 0;
    int cols = 0;
    size_t slot_bytes = 0, data_bytes = 0;
    char *data = nullptr;            // own data window
    FlagWin *flags = nullptr;        // own flag window
    Local *loc = nullptr;
    float *staging[2] = {nullptr, nullptr};
    char *peer_data[QS_XCHG_MAX_RANKS] = {};
    FlagWin *peer_flags[QS_XCHG_MAX_RANKS] = {};
    bool opened[QS_XCHG_MAX_RANKS] = {};   // mapped with hipIpcOpenMemHandle (to be closed)
    XchgDev *desc = nullptr;               // device copy of what a step kernel needs for the fused push (qs_xchg_fused_desc)
    unsigned long long timeout_ticks = 0;
    int fenced = 0;                        // created under QS_XCHG_FENCED=1: the fenced variant of the flag protocol (qs_xchg_dev.h)
    Q8Dev q8 = {};                         // wire == QS_WIRE_Q8
    size_t rank_bytes = 0;                 // bytes of one rank's rows in a window slot
};

// device-side form of a QS_WIRE_Q8 layout; false if the layout is not one
static bool q8_dev(int32_t cols, const qs_wire_q8 *l, Q8Dev &q) {
    if (!l || cols < 1 || l->q0 < 0 || l->q1 < l->q0 || l->q1 > cols) return false;
    memset(&q, 0, sizeof q);
    q.D = cols; q.q0 = l->q0; q.q1 = l->q1;
    const int c16 = cols - (l->q1 - l->q0), n8 = l->q1 - l->q0;
    q.w16 = (c16 + 1) / 2;
    q.row_words = q.w16 + (n8 + 3) / 4;
    float sc[6];
    for (int a = 0; a < 6; ++a) { if (!(l->clip[a] > 0.0f)) return false; sc[a] = (float)(127.0 / (double)l->clip[a]); }
    q.s0 = sc[0]; q.s1 = sc[1]; q.s2 = sc[2]; q.s3 = sc[3]; q.s4 = sc[4]; q.s5 = sc[5];
    return true;
}

extern "C" {

const char *qs_xchg_last_error(void) { return g_err.c_str(); }

int64_t qs_wire_row_bytes(int32_t cols, int wire, const qs_wire_q8 *layout) {
    if (wire == QS_WIRE_F32) return 4LL * cols;
    if (wire == QS_WIRE_BF16) return 2LL * cols;
    Q8Dev q;
    if (wire != QS_WIRE_Q8 || !q8_dev(cols, layout, q)) return -1;
    return 4LL * q.row_words;
}

static int xchg_create(int device, int world, int rank, int64_t rows, int32_t cols, int wire, const qs_wire_q8 *layout, qs_xchg **out);
int qs_xchg_create(int device, int world, int rank, int64_t rows, int32_t cols, int wire, qs_xchg **out) {
    if (wire != QS_WIRE_F32 && wire != QS_WIRE_BF16) return fail(-1, "qs_xchg_create: bad argument (QS_WIRE_Q8: qs_xchg_create_q8)");
    return xchg_create(device, world, rank, rows, cols, wire, nullptr, out);
}
int qs_xchg_create_q8(int device, int world, int rank, int64_t rows, int32_t cols, const qs_wire_q8 *layout, qs_xchg **out) {
    Q8Dev q;
    if (!q8_dev(cols, layout, q)) return fail(-1, "qs_xchg_create_q8: bad layout");
    return xchg_create(device, world, rank, rows, cols, QS_WIRE_Q8, layout, out);
}
static int xchg_create(int device, int world, int rank, int64_t rows, int32_t cols, int wire, const qs_wire_q8 *layout, qs_xchg **out) {
    if (!out || world < 1 || world > QS_XCHG_MAX_RANKS || rank < 0 || rank >= world || rows < 1 || cols < 1)
        return fail(-1, "qs_xchg_create: bad argument");
    int ndev = 0;
    XTRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(-2, "no such HIP device");
    XTRY(hipSetDevice(device));
    qs_xchg *x = new qs_xchg();
    x->device = device; x->world = world; x->rank = rank; x->wire = wire; x->rows = rows; x->cols = cols; x->n = (long long)rows * cols;
    if (wire == QS_WIRE_Q8) q8_dev(cols, layout, x->q8);
    x->rank_bytes = (size_t)rows * (size_t)qs_wire_row_bytes(cols, wire, layout);
    x->slot_bytes = ((size_t)world * x->rank_bytes + 255) & ~(size_t)255;
    x->data_bytes = 2 * x->slot_bytes;
    hipError_t e = hipMalloc((void **)&x->data, x->data_bytes);
    // flags: fine-grained (uncached) device memory when the runtime offers it - remote stores to them are polled, not read once
    if (e == hipSuccess) {
        e = hipExtMallocWithFlags((void **)&x->flags, sizeof(FlagWin), hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc((void **)&x->flags, sizeof(FlagWin)); }
    }
    if (e == hipSuccess) e = hipMalloc((void **)&x->loc, sizeof(Local));
    const size_t stage_bytes = ((size_t)x->n * 4 + 255) & ~(size_t)255;
    if (e == hipSuccess) e = hipMalloc((void **)&x->staging[0], 2 * stage_bytes);
    if (e != hipSuccess) { qs_xchg_destroy(x); return fail(-2, std::string("qs_xchg_create: ") + hipGetErrorString(e)); }
    x->staging[1] = (float *)((char *)x->staging[0] + stage_bytes);
    XTRY(hipMemset(x->data, 0, x->data_bytes));
    XTRY(hipMemset(x->flags, 0, sizeof(FlagWin)));
    XTRY(hipMemset(x->loc, 0, sizeof(Local)));
    XTRY(hipMemset(x->staging[0], 0, 2 * stage_bytes));
    XTRY(hipDeviceSynchronize());
    int khz = 100000;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) { (void)hipGetLastError();
        khz = 100000; }
    x->timeout_ticks = (unsigned long long)khz * QS_XCHG_TIMEOUT_MS;
    if (const char *ev = getenv("QS_XCHG_TIMEOUT_MS")) { const long v = atol(ev);
        if (v > 0) x->timeout_ticks = (unsigned long long)khz * (unsigned long long)v; }
    if (const char *ev = getenv("QS_XCHG_FENCED")) x->fenced = atoi(ev) != 0 ? 1 : 0;
    x->peer_data[rank] = x->data;
    x->peer_flags[rank] = x->flags;
    *out = x;
    return 0;
}

// the flag protocol of this endpoint (quadswarm_exchange.h): every launch reads x->fenced when it is enqueued
int qs_xchg_set_fenced(qs_xchg *x, int fenced) {
    if (!x) return fail(-1, "qs_xchg_set_fenced: null endpoint");
    x->fenced = fenced ? 1 : 0;
    return 0;
}
int qs_xchg_get_fenced(qs_xchg *x) { return x ? x->fenced : -1; }

int qs_xchg_destroy(qs_xchg *x) {
    if (!x) return 0;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (x->opened[r]) { (void)hipIpcCloseMemHandle(x->peer_data[r]); (void)hipIpcCloseMemHandle(x->peer_flags[r]); }
    if (x->data) (void)hipFree(x->data);
    if (x->flags) (void)hipFree(x->flags);
    if (x->loc) (void)hipFree(x->loc);
    if (x->staging[0]) (void)hipFree(x->staging[0]);
    if (x->desc) (void)hipFree(x->desc);
    delete x;
    return 0;
}

int qs_xchg_export(qs_xchg *x, void *blob_out) {
    if (!x || !blob_out) return fail(-1, "null argument");
    XTRY(hipSetDevice(x->device));
    Blob b;
    memset(&b, 0, sizeof b);
    XTRY(hipIpcGetMemHandle(&b.data, x->data));
    XTRY(hipIpcGetMemHandle(&b.flags, x->flags));
    b.pid = (int32_t)getpid(); b.device = x->device;
    memcpy(blob_out, &b, sizeof b);
    return 0;
}

int qs_xchg_attach(qs_xchg *x, const void *blobs) {
    if (!x || !blobs) return fail(-1, "null argument");
    XTRY(hipSetDevice(x->device));
    const Blob *bl = (const Blob *)blobs;
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        if (bl[r].pid == (int32_t)getpid()) return fail(-1, "qs_xchg_attach: peer lives in this process, use qs_xchg_attach_local");
        if (bl[r].device != x->device) {   // direct stores over xGMI need peer access; already-enabled is fine
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, x->device, bl[r].device) == hipSuccess && can) {
                hipError_t pe = hipDeviceEnablePeerAccess(bl[r].device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();
            } else (void)hipGetLastError();
        }
        void *pd = nullptr, *pf = nullptr;
        XTRY(hipIpcOpenMemHandle(&pd, bl[r].data, hipIpcMemLazyEnablePeerAccess));
        XTRY(hipIpcOpenMemHandle(&pf, bl[r].flags, hipIpcMemLazyEnablePeerAccess));
        x->peer_data[r] = (char *)pd; x->peer_flags[r] = (FlagWin *)pf; x->opened[r] = true;
    }
    return 0;
}

int qs_xchg_attach_local(qs_xchg *x, int peer_rank, qs_xchg *peer) {
    if (!x || !peer || peer_rank < 0 || peer_rank >= x->world || peer_rank == x->rank) return fail(-1,
        "qs_xchg_attach_local: bad argument");
    if (peer->world != x->world || peer->n != x->n || peer->wire != x->wire || peer->rank_bytes != x->rank_bytes
        || peer->rank != peer_rank) return fail(-1, "qs_xchg_attach_local: endpoints do not match");
    if (peer->device != x->device) {
        XTRY(hipSetDevice(x->device));
        hipError_t pe = hipDeviceEnablePeerAccess(peer->device, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(-2,
            std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
        (void)hipGetLastError();
    }
    x->peer_data[peer_rank] = peer->data; x->peer_flags[peer_rank] = peer->flags;
    return 0;
}

void *qs_xchg_staging(qs_xchg *x, int slot) { return x ? (void *)x->staging[slot & 1] : nullptr; }
void *qs_xchg_gathered(qs_xchg *x, int slot) { return x ? (void *)(x->data + (size_t)(slot & 1) * x->slot_bytes) : nullptr; }

static int check_wired(qs_xchg *x) {
    for (int r = 0; r < x->world; ++r)
        if (!x->peer_data[r] || !x->peer_flags[r]) return fail(-1, "exchange endpoint not attached to all peers");
    return 0;
}

int qs_xchg_push(qs_xchg *x, const void *src_f32, void *stream) {
    if (!x) return fail(-1, "null endpoint");
    if (int rc = check_wired(x)) return rc;
    XTRY(hipSetDevice(x->device));
    PushArgs a;
    memset(&a, 0, sizeof a);
    a.src = (const float *)src_f32; a.staging[0] = x->staging[0]; a.staging[1] = x->staging[1];
    for (int r = 0; r < x->world; ++r) { a.data_win[r] = x->peer_data[r]; a.flag_win[r] = x->peer_flags[r]; }
    a.mine = x->flags; a.loc = x->loc; a.n = x->n; a.slot_bytes = (long long)x->slot_bytes; a.world = x->world; a.rank = x->rank;
    a.wire = x->wire;
    a.timeout_ticks = x->timeout_ticks; a.q8 = x->q8; a.rows = x->rows; a.fenced = x->fenced;
    hipLaunchKernelGGL(qs_xchg_push_kernel, dim3(grid_parts(x->n), x->world), dim3(256), 0, (hipStream_t)stream, a);
    XTRY(hipGetLastError());
    return 0;
}

static void release_args(qs_xchg *x, ReleaseArgs &a) {
    memset(&a, 0, sizeof a);
    for (int r = 0; r < x->world; ++r) a.flag_win[r] = x->peer_flags[r];
    a.loc = x->loc; a.world = x->world; a.rank = x->rank; a.fenced = x->fenced;
}

static int launch_wait(qs_xchg *x, void *stream, int release) {
    if (!x) return fail(-1, "null endpoint");
    if (release) { if (int rc = check_wired(x)) return rc; }
    XTRY(hipSetDevice(x->device));
    ReleaseArgs a;
    release_args(x, a);
    hipLaunchKernelGGL(qs_xchg_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x->flags, a, x->timeout_ticks, release);
    XTRY(hipGetLastError());
    return 0;
}
int qs_xchg_wait(qs_xchg *x, void *stream) { return launch_wait(x, stream, 0); }
int qs_xchg_wait_release(qs_xchg *x, void *stream) { return launch_wait(x, stream, 1); }

int qs_xchg_release(qs_xchg *x, void *stream) {
    if (!x) return fail(-1, "null endpoint");
    if (int rc = check_wired(x)) return rc;
    XTRY(hipSetDevice(x->device));
    ReleaseArgs a;
    release_args(x, a);
    hipLaunchKernelGGL(qs_xchg_release_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    XTRY(hipGetLastError());
    return 0;
}

// Device-resident descriptor for the FUSED form (quadswarm.h: qs_set_obs_exchange): the team step kernels store their rows into every
// rank's window themselves.  blocks = workgroups of one step launch; auto_ack: that launch also waits for the rows of all ranks and
// releases the slot.  Returns the device pointer (owned by the endpoint), or NULL (qs_xchg_last_error()).
void *qs_xchg_fused_desc(qs_xchg *x, int32_t blocks, int32_t auto_ack, int64_t *n_out) {
    if (!x || blocks < 1) { g_err = "qs_xchg_fused_desc: bad argument"; return nullptr; }
    if (check_wired(x)) return nullptr;
    if (hipSetDevice(x->device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    XchgDev d;
    memset(&d, 0, sizeof d);
    for (int r = 0; r < x->world; ++r) { d.data_win[r] = x->peer_data[r]; d.flag_win[r] = x->peer_flags[r]; }
    d.mine = x->flags; d.loc = x->loc; d.n = x->n; d.slot_bytes = (long long)x->slot_bytes; d.world = x->world; d.rank = x->rank;
    d.wire = x->wire;
    d.auto_ack = auto_ack ? 1 : 0; d.blocks = (unsigned int)blocks; d.timeout_ticks = x->timeout_ticks; d.q8 = x->q8; d.fenced = x->fenced;
    if (!x->desc && hipMalloc((void **)&x->desc, sizeof d) != hipSuccess) { g_err = "hipMalloc failed"; return nullptr; }
    if (hipDeviceSynchronize() != hipSuccess
        || hipMemcpy(x->desc, &d, sizeof d, hipMemcpyHostToDevice) != hipSuccess) { g_err = "descriptor upload failed"; return nullptr; }
    if (n_out) *n_out = x->n;
    return x->desc;
}

// library-internal (qs_set_obs_exchange): does the endpoint carry `cols`-column rows in wire `wire` and, for QS_WIRE_Q8, the block [q0,
// q1)?
int qs_xchg_row_layout_is(qs_xchg *x, int32_t cols, int32_t q0, int32_t q1) {
    if (!x) return 0;
    if (x->cols != cols) return 0;
    if (x->wire != QS_WIRE_Q8) return 1;
    return (x->q8.D == cols && x->q8.q0 == q0 && x->q8.q1 == q1) ? 1 : 0;
}

int qs_xchg_status(qs_xchg *x, int64_t out[4]) {
    if (!x || !out) return fail(-1, "null argument");
    XTRY(hipSetDevice(x->device));
    Local l;
    XTRY(hipMemcpy(&l, x->loc, sizeof l, hipMemcpyDeviceToHost));
    out[0] = l.status; out[1] = (int64_t)l.push_seq; out[2] = (int64_t)l.wait_seq; out[3] = (int64_t)l.release_seq;
    return 0;
}

int qs_obs_pack(const void *src_f32, void *dst, int64_t n, int wire, void *stream) {
    if (!src_f32 || !dst || n < 0 || (wire != QS_WIRE_F32 && wire != QS_WIRE_BF16)) return fail(-1, "qs_obs_pack: bad argument");
    if (n == 0) return 0;
    int parts = (int)(((n >> 3) + 1023) / 1024);
    parts = parts < 1 ? 1 : (parts > 1024 ? 1024 : parts);
    hipLaunchKernelGGL(qs_obs_pack_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, (const float *)src_f32, (char *)dst,
        (long long)n, wire);
    XTRY(hipGetLastError());
    return 0;
}

int qs_obs_pack_rows(const void *src_f32, void *dst, int64_t rows, int32_t cols, int wire, const qs_wire_q8 *layout, void *stream) {
    if (wire != QS_WIRE_Q8) return qs_obs_pack(src_f32, dst, rows * (int64_t)cols, wire, stream);
    Q8Dev q;
    if (!src_f32 || !dst || rows < 0 || !q8_dev(cols, layout, q)) return fail(-1, "qs_obs_pack_rows: bad argument");
    if (rows == 0) return 0;
    int parts = (int)((rows * q.row_words / 4 + 1023) / 1024);
    parts = parts < 1 ? 1 : (parts > 1024 ? 1024 : parts);
    hipLaunchKernelGGL(qs_obs_pack_q8_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, (const float *)src_f32, (char *)dst,
        (long long)rows, q);
    XTRY(hipGetLastError());
    return 0;
}

int qs_obs_unpack_rows(const void *src_wire, void *dst_f32, int64_t rows, int32_t cols, int wire, const qs_wire_q8 *layout, void *stream) {
    Q8Dev q;
    memset(&q, 0, sizeof q);
    if (!src_wire || !dst_f32 || rows < 0 || cols < 1 || (wire != QS_WIRE_F32 && wire != QS_WIRE_BF16 && wire != QS_WIRE_Q8)
        || (wire == QS_WIRE_Q8 && !q8_dev(cols, layout, q)))
        return fail(-1, "qs_obs_unpack_rows: bad argument");
    if (rows == 0) return 0;
    long long parts = (rows * cols + 2047) / 2048;
    parts = parts < 1 ? 1 : (parts > 2048 ? 2048 : parts);
    hipLaunchKernelGGL(qs_obs_unpack_kernel, dim3((int)parts), dim3(256), 0, (hipStream_t)stream, (const char *)src_wire,
        (float *)dst_f32, (long long)rows, (int)cols, wire, q);
    XTRY(hipGetLastError());
    return 0;
}

}   // extern "C"
