// One-shot gradient all-reduce over xGMI peer-to-peer writes (replaces the RCCL all-reduce of the DQN's 724 KB
// gradient arena in the data-parallel step when the ranks of a node can map each other's memory).
//
// xGMI is point-to-point: every GPU has a direct link to each of the 7 others.  A ring all-reduce of a 724 KB message is
// 2 (N-1) dependent hops of latency; here every rank PUSHES its gradient into its own slot of every peer's arena (7 links in
// parallel, one hop), raises a flag there, and then sums the N slots it received in rank order — the same order on every
// rank, so replicas stay bit-identical.  Two kernels per step and rank, no host synchronisation:
//
//   push    grad -> slot[rank] of every arena (own included) ; system-scope fence per workgroup ; the LAST workgroup
//           (device-scope ticket) stores the step's epoch into flag[rank] of every arena (system-scope release)
//   reduce  every workgroup waits until flag[s] == epoch for all s (system-scope acquire, bounded spin), then
//           out[i] = slot[0][i] + slot[1][i] + ... ; a timeout sets an error word instead of hanging the GPU
//
// The arena is fine-grained device memory (peer writes and local reads must be coherent INSIDE kernels, which coarse-grained
// memory only promises at kernel boundaries) — the one device allocation this library makes itself, because neither torch
// nor a caller can allocate that kind: ivosw_p2p_alloc / ivosw_p2p_free.  Slots are double-buffered by epoch parity: a
// rank pushes epoch k+2 only after its reduce of k+1, which needed every peer's flag k+1, i.e. every peer had finished
// reading epoch k.  Peers are mapped with hipIpcOpenMemHandle (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0).
// The host side (ivos_w_amd/parallel.py) self-tests the path against torch.distributed's all-reduce at start-up and every
// rank falls back to RCCL unless all of them pass.
#include <algorithm>

#include "adam.h"

namespace ivosw {

struct P2pHeader {                 // at the start of every arena
    unsigned flags[2][64];         // [parity][source rank]: last epoch whose data from that rank is complete
    unsigned ticket;               // workgroups of the running push that are done (0 between launches)
    unsigned error;                // set by reduce on timeout
    unsigned verdict;              // (epoch << 2) | 1 = every workgroup applies this epoch, | 2 = none does: decided ONCE per epoch by compare-and-swap
    unsigned pad[61];
};
static_assert(sizeof(P2pHeader) % 256 == 0, "slots stay 256-byte aligned");
constexpr int P2P_MAX_WORLD = 64;

__host__ __device__ inline size_t p2p_slot_floats(size_t n) { return (n + 63) / 64 * 64; }
__host__ __device__ inline float* p2p_slot(void* arena, int world, size_t n, int parity, int src) {
    return reinterpret_cast<float*>(static_cast<char*>(arena) + sizeof(P2pHeader)) + ((size_t)parity * world + src) * p2p_slot_floats(n);
}

struct P2pPeers { void* arena[P2P_MAX_WORLD]; };

__global__ __launch_bounds__(1024) void p2p_push_kernel(const float* __restrict__ grad, int n, int rank, int world, P2pPeers peers, unsigned epoch) {
    const int parity = epoch & 1;
    const int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 < n) {
        float4 v;
        if (i4 + 3 < n) v = *reinterpret_cast<const float4*>(grad + i4);
        else { v.x = grad[i4]; v.y = i4 + 1 < n ? grad[i4 + 1] : 0.f; v.z = i4 + 2 < n ? grad[i4 + 2] : 0.f; v.w = 0.f; }
        for (int d = 0; d < world; ++d) {                    // start with the next rank: the 7 links are used at the same time
            const int dst = (rank + 1 + d) % world;
            *reinterpret_cast<float4*>(p2p_slot(peers.arena[dst], world, n, parity, rank) + i4) = v;      // slots are padded to 64 floats
        }
    }
    __threadfence_system();                                  // this workgroup's remote writes are visible before its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        P2pHeader* own = static_cast<P2pHeader*>(peers.arena[rank]);
        if (atomicAdd(&own->ticket, 1u) == gridDim.x - 1) {
            atomicExch(&own->ticket, 0u);
            __threadfence_system();
            for (int d = 0; d < world; ++d) {
                P2pHeader* h = static_cast<P2pHeader*>(peers.arena[(rank + 1 + d) % world]);
                __hip_atomic_store(&h->flags[parity][rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Wait for the `world` flags of this epoch.  The verdict is GRID-WIDE and decided ONCE per epoch through one word (ADVICE round 4:
// the round-3 form - workgroup 0 alone declares a timeout through the error word - left a window between its lane's deadline and
// the word's publication in which a late flag let other workgroups apply their slice of clamp + Adam): a workgroup that saw every
// flag proposes (epoch, OK), a workgroup whose deadline passed proposes (epoch, TIMEOUT), both by atomicCAS on h->verdict, and EVERY
// workgroup acts on the value that is in the word afterwards - the first proposal wins for the whole grid, so either every workgroup
// applies its slice or none does.  Deadlines: workgroup 0 after the timeout, the others after twice that (should workgroup 0 never
// run).  A TIMEOUT verdict also sets the sticky error word the host reads.
__device__ __forceinline__ bool p2p_wait_all(P2pHeader* h, int world, int parity, unsigned epoch, unsigned long long timeout_ticks) {
    __shared__ int state;                                    // 0 undecided here, 1 all flags seen, 2 timed out / verdict already there
    if (threadIdx.x == 0) state = 1;
    __syncthreads();
    if ((int)threadIdx.x < world) {                          // lane s waits for rank s
        const unsigned long long t0 = wall_clock64();
        const unsigned long long limit = blockIdx.x == 0 ? timeout_ticks : 2 * timeout_ticks;
        while (__hip_atomic_load(&h->flags[parity][threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
            const unsigned w = __hip_atomic_load(&h->verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (wall_clock64() - t0 > limit || (w >> 2) == epoch) { state = 2; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned mine = (epoch << 2) | (state == 1 ? 1u : 2u);
        // the verdict word is published and read with release / acquire: a workgroup that adopts an OK verdict it did not
        // propose itself synchronises with the proposer, who acquire-loaded every flag before proposing
        unsigned w = __hip_atomic_load(&h->verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        while ((w >> 2) != epoch) {                          // undecided for this epoch: propose; whoever swaps first has decided
            unsigned expect = w;
            if (__hip_atomic_compare_exchange_strong(&h->verdict, &expect, mine, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) { w = mine; break; }
            w = expect;
        }
        state = (w & 3u) == 1u ? 1 : 2;
        if (state == 2) atomicExch(&h->error, 1u);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");            // system scope: nothing cached from the slots' previous contents
    return state == 1;
}

__global__ __launch_bounds__(1024) void p2p_reduce_kernel(float* __restrict__ out, int n, int world, void* arena, unsigned epoch,
                                                          unsigned long long timeout_ticks) {
    P2pHeader* h = static_cast<P2pHeader*>(arena);
    const int parity = epoch & 1;
    if (!p2p_wait_all(h, world, parity, epoch, timeout_ticks)) return;
    const int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    float4 s = *reinterpret_cast<const float4*>(p2p_slot(arena, world, n, parity, 0) + i4);
    for (int r = 1; r < world; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(p2p_slot(arena, world, n, parity, r) + i4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (i4 + 3 < n) *reinterpret_cast<float4*>(out + i4) = s;
    else { out[i4] = s.x; if (i4 + 1 < n) out[i4 + 1] = s.y; if (i4 + 2 < n) out[i4 + 2] = s.z; }
}

// The same wait + rank-ordered sum, with clamp + Adam applied to the sum as it is formed ("every GPU sums 8 slots
// locally inside the clamp_adam kernel"): the summed gradient never makes a round trip through HBM and the data-parallel step
// loses a launch.  gout (may be NULL) receives the summed, unscaled gradient.  On a timeout NOTHING is updated (the verdict is grid-wide: p2p_wait_all) and the error word
// is set: the host raises (ivos_w_amd.parallel), replicas cannot drift apart silently.
__global__ __launch_bounds__(1024) void p2p_reduce_clamp_adam_kernel(float* __restrict__ gout, int n, int world, void* arena, unsigned epoch,
                                                                     unsigned long long timeout_ticks, float* __restrict__ p,
                                                                     float* __restrict__ m, float* __restrict__ v, float step_size,
                                                                     float bc2_sqrt, float beta1, float beta2, float eps, float wd,
                                                                     float clampv, float gscale) {
    P2pHeader* h = static_cast<P2pHeader*>(arena);
    const int parity = epoch & 1;
    if (!p2p_wait_all(h, world, parity, epoch, timeout_ticks)) return;
    const int i4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    float4 s = *reinterpret_cast<const float4*>(p2p_slot(arena, world, n, parity, 0) + i4);
    for (int r = 1; r < world; ++r) {
        const float4 q = *reinterpret_cast<const float4*>(p2p_slot(arena, world, n, parity, r) + i4);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = i4 + j;
        if (i < n) {
            float mi = m[i], vi = v[i];
            p[i] = clamp_adam_elem(sv[j], p[i], mi, vi, step_size, bc2_sqrt, beta1, beta2, eps, wd, clampv, gscale);
            m[i] = mi;
            v[i] = vi;
            if (gout) gout[i] = sv[j];
        }
    }
}

}  // namespace ivosw

using namespace ivosw;

// ivosw_p2p_allreduce + ivosw_clamp_adam(..., grad_scale = 1/world) as TWO launches instead of three: push, then wait + sum + clamp +
// Adam.  grads_out (may be NULL, may alias grads) receives the summed gradient.  `step` is the 1-based Adam step, as in ivosw_clamp_adam.
extern "C" int ivosw_p2p_allreduce_clamp_adam(const float* grads, float* grads_out, int n, int rank, int world, void* const* arenas,
                                              unsigned epoch, int timeout_ms, float* params, float* exp_avg, float* exp_avg_sq, int step,
                                              float lr, float beta1, float beta2, float eps, float weight_decay, float clamp,
                                              ivosw_stream_t stream) {
    IVOSW_REQUIRE(grads && arenas && params && exp_avg && exp_avg_sq, "null pointer");
    IVOSW_ON_DEVICE_OF(params);
    IVOSW_REQUIRE(n > 0 && world > 0 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world && epoch > 0 && step >= 1, "bad rank / world / epoch / step");
    IVOSW_REQUIRE(epoch < (1u << 30), "epoch must stay below 2^30: the verdict word packs (epoch << 2) | decision");
    IVOSW_REQUIRE((reinterpret_cast<uintptr_t>(grads) & 15) == 0, "the gradient buffer must be 16-byte aligned");
    P2pPeers peers{};
    for (int r = 0; r < world; ++r) {
        IVOSW_REQUIRE(arenas[r], "null arena");
        peers.arena[r] = arenas[r];
    }
    const double bc1 = 1.0 - ipow((double)beta1, step), bc2 = 1.0 - ipow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipStream_t st = as_stream(stream);
    const int nblk = ((n + 3) / 4 + 1023) / 1024;
    hipLaunchKernelGGL(p2p_push_kernel, dim3(nblk), dim3(1024), 0, st, grads, n, rank, world, peers, epoch);
    hipLaunchKernelGGL(p2p_reduce_clamp_adam_kernel, dim3(nblk), dim3(1024), 0, st, grads_out, n, world, arenas[rank], epoch,
                       (unsigned long long)std::max(1, timeout_ms) * 100000ull, params, exp_avg, exp_avg_sq, step_size, bc2_sqrt, beta1, beta2,
                       eps, weight_decay, clamp, 1.0f / (float)world);
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}

extern "C" size_t ivosw_p2p_arena_bytes(int world, size_t n) {
    if (world <= 0 || world > P2P_MAX_WORLD || n == 0) return 0;
    return sizeof(P2pHeader) + (size_t)2 * world * p2p_slot_floats(n) * sizeof(float);
}

extern "C" int ivosw_p2p_alloc(size_t bytes, void** arena, void* ipc_handle, size_t ipc_handle_bytes) {
    IVOSW_REQUIRE(arena && ipc_handle && bytes >= sizeof(P2pHeader), "null pointer / arena too small");
    IVOSW_REQUIRE(ipc_handle_bytes >= sizeof(hipIpcMemHandle_t), "ipc handle buffer too small (ivosw_p2p_handle_bytes)");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("ivosw_p2p_alloc: hipExtMallocWithFlags: %s", hipGetErrorString(e)); return IVOSW_ERR_LAUNCH; }
    e = hipMemset(p, 0, bytes);
    hipIpcMemHandle_t hnd;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hnd, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        set_error("ivosw_p2p_alloc: %s", hipGetErrorString(e));
        return IVOSW_ERR_LAUNCH;
    }
    memcpy(ipc_handle, &hnd, sizeof(hnd));
    *arena = p;
    return IVOSW_OK;
}

extern "C" size_t ivosw_p2p_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int ivosw_p2p_open(const void* ipc_handle, void** peer_arena) {
    IVOSW_REQUIRE(ipc_handle && peer_arena, "null pointer");
    hipIpcMemHandle_t hnd;
    memcpy(&hnd, ipc_handle, sizeof(hnd));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, hnd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("ivosw_p2p_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return IVOSW_ERR_LAUNCH; }
    *peer_arena = p;
    return IVOSW_OK;
}

extern "C" int ivosw_p2p_close(void* peer_arena) {
    if (peer_arena && hipIpcCloseMemHandle(peer_arena) != hipSuccess) (void)hipGetLastError();
    return IVOSW_OK;
}

extern "C" int ivosw_p2p_free(void* arena) {
    if (arena && hipFree(arena) != hipSuccess) (void)hipGetLastError();
    return IVOSW_OK;
}

extern "C" int ivosw_p2p_error(const void* arena, int* error) {
    IVOSW_REQUIRE(arena && error, "null pointer");
    unsigned v = 0;
    hipError_t e = hipMemcpy(&v, &static_cast<const P2pHeader*>(arena)->error, sizeof(v), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("ivosw_p2p_error: %s", hipGetErrorString(e)); return IVOSW_ERR_LAUNCH; }
    *error = (int)v;
    return IVOSW_OK;
}

// grads [n] of this rank -> summed over the `world` ranks into out [n] (may alias grads).  arenas: HOST array of `world`
// device pointers, arenas[rank] = own arena, the others = ivosw_p2p_open of the peers'; epoch: 1, 2, 3, ... the same on
// every rank (one per call).  timeout_ms bounds the wait for the peers' flags (then the arena's error word is set).
extern "C" int ivosw_p2p_allreduce(const float* grads, float* out, int n, int rank, int world, void* const* arenas, unsigned epoch,
                                   int timeout_ms, ivosw_stream_t stream) {
    IVOSW_REQUIRE(grads && out && arenas, "null pointer");
    IVOSW_ON_DEVICE_OF(out);
    IVOSW_REQUIRE(n > 0 && world > 0 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world && epoch > 0, "bad rank / world / epoch");
    IVOSW_REQUIRE(epoch < (1u << 30), "epoch must stay below 2^30: the verdict word packs (epoch << 2) | decision");
    IVOSW_REQUIRE(((reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "gradient buffers must be 16-byte aligned");
    P2pPeers peers{};
    for (int r = 0; r < world; ++r) {
        IVOSW_REQUIRE(arenas[r], "null arena");
        peers.arena[r] = arenas[r];
    }
    hipStream_t st = as_stream(stream);
    const int nblk = ((n + 3) / 4 + 1023) / 1024;
    hipLaunchKernelGGL(p2p_push_kernel, dim3(nblk), dim3(1024), 0, st, grads, n, rank, world, peers, epoch);
    hipLaunchKernelGGL(p2p_reduce_kernel, dim3(nblk), dim3(1024), 0, st, out, n, world, arenas[rank], epoch,
                       (unsigned long long)std::max(1, timeout_ms) * 100000ull);       // wall_clock64: 100 MHz
    IVOSW_CHECK_LAUNCH();
    return IVOSW_OK;
}
