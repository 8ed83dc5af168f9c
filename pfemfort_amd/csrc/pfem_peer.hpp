// pfem_peer.hpp -- device kernels of the peer-memory transport (pfem_solver_set_comm_peer): neighbour exchange and small
// all-reduces through receive buffers that the ranks map into each other's address space (hipIpcGetMemHandle /
// hipIpcOpenMemHandle), with one release/acquire flag per (rank, neighbour) pair -- what the VecScatter inside KSPSolve does
// for the reference (solverpetsc.F:476), without a collective library and without the host in the data path.  SURVEY 8(e):
// the messages are 62 KB ... 1.3 MB faces and a few scalars, latency-bound; a direct write + flag is the shortest path.
//
// Visibility (cdna_hip_programming.md G16): payload stores -> system-scope fence -> flag store (release); the consumer polls
// the flag with acquire loads, fences, then reads the payload.  Per-XCD L2s are not coherent with each other, so both
// fences are needed even between two processes on ONE device.  Every wait is bounded (kPeerSpinTicks of the 100 MHz
// wall clock); a timeout sets the error word the host reads after the solve instead of hanging the device -- and, so that the
// solve does not run on with stale data at 10 s per exchange: the CG's control word (abort_word: every kernel of the iteration
// leaves at once when it is non-zero) and every later exchange / all-reduce of this rank returns at once (the error word is
// sticky; the transport is finished, the solve reports PFEM_ERR_COMM).
// ORDER: all exchanges of a pair of ranks must be enqueued in one order on both (the solver enqueues them on its compute
// stream or on its communication stream behind an event, never concurrently): message e + 1 of a pair may land while message e
// is still unread -- it goes to the box of the other parity -- but nothing beyond that; a flag further ahead is a protocol
// error and is reported as one, not read past.
#pragma once

namespace pfem {

constexpr unsigned long long kPeerSpinTicks = 1000000000ull;      // 10 s at 100 MHz

struct PeerMail {                 // one rank's mailbox as a peer sees it (all device pointers, mapped into THIS process)
    unsigned long long *xflag;    // [nranks] epoch of the last exchange message that has landed, by source rank
    unsigned long long *xack;     // [nranks] epoch of MY message to that rank that it has consumed (written by that rank)
    unsigned long long *aflag;    // [nranks] all-reduce epoch that has landed, by source rank
    unsigned long long *aack;     // [nranks] all-reduce epoch that rank has consumed
    int *err;                     // timeout / protocol error
    double *xbox;                 // [2][nranks][cap_x] exchange payload by epoch parity and source rank
    double *abox;                 // [2][nranks][cap_a] all-reduce payload
};
constexpr int kPeerMaxRanks = 16;
constexpr int kPeerAbortReason = -100;     // what the CG's control word says after a transport failure (not a KSPConvergedReason)
struct PeerWorld {
    PeerMail m[kPeerMaxRanks];    // m[rank] = this rank's own mailbox (local pointers)
    int rank, nranks;
    int64_t cap_x, cap_a;
    int *abort_word;              // the running solve's control word (CgCtl::flag) or null
    // message counters live on the DEVICE (round 5): a launch carries no epoch, so the same launch can be replayed from a
    // captured hipGraph -- the multigrid cycle across ranks enqueues ~65 kernels per iteration, which the host otherwise
    // re-issues one by one.  xepoch[q]: exchange messages sent to rank q so far; aepoch: all-reduces so far.
    unsigned long long *xepoch, *aepoch;
};
__device__ __forceinline__ void peer_fail(const PeerWorld &W)
{
    *W.m[W.rank].err = 1;
    if (W.abort_word) __hip_atomic_store(W.abort_word, kPeerAbortReason, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until *flag >= want; false (and the failure recorded) on a timeout or when the flag is more than `ahead` beyond it
__device__ __forceinline__ bool peer_wait(const PeerWorld &W, const unsigned long long *flag, unsigned long long want, unsigned long long ahead)
{
    const unsigned long long t0 = wall_clock64();
    unsigned long long v;
    while ((v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < want) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > kPeerSpinTicks) { peer_fail(W); return false; }
    }
    if (v > want + ahead) { peer_fail(W); return false; }       // the ranks' message counts have come apart
    return true;
}

// kPeerXBlocks blocks per neighbour (a segment is cut into equal slices; round 4 had ONE block push a whole face: 111 us for
// the 1.29 MB faces of config 5): every block waits for the slot, writes its slice of my segment into the neighbour's box; the
// LAST block to finish (arrival counter) fences and raises the flag; every block waits for the neighbour's flag, copies its slice
// of the box out; the last one acknowledges and advances the pair's message counter.  No block waits for another block of
// this launch -- only for the neighbour -- so the blocks need not be co-resident.
constexpr int kPeerXBlocks = 16;
constexpr int64_t kPeerXSlice = 8192;             // doubles per block at least (64 KB): small faces keep one block
struct PeerExchangeArgs {
    int np;
    int peers[kPeerMaxRanks];
    int64_t off[kPeerMaxRanks + 1];
    int blocks[kPeerMaxRanks];                    // blocks that serve neighbour k (1 .. kPeerXBlocks); block index = kPeerXBlocks * k + b
};
__global__ void __launch_bounds__(1024) k_peer_exchange(PeerWorld W, PeerExchangeArgs X, const double *__restrict__ send, double *__restrict__ recv,
                                                        int *__restrict__ arrive /* [2][kPeerMaxRanks], zero between launches */)
{
    __shared__ int ok_s, last_s;
    const int k = blockIdx.x / kPeerXBlocks, b = blockIdx.x % kPeerXBlocks;
    if (k >= X.np || b >= X.blocks[k]) return;
    const int q = X.peers[k], r = W.rank, nb = X.blocks[k];
    // this pair's message number (starts at 1): advanced by the last block of this launch, and the launches of a pair are ordered
    const unsigned long long e = W.xepoch[q] + 1;
    const int64_t cnt = X.off[k + 1] - X.off[k];
    const int64_t per = (cnt + nb - 1) / nb, i0 = per * b, i1 = i0 + per < cnt ? i0 + per : cnt;
    const PeerMail &mine = W.m[r], &his = W.m[q];
    // (a transport that has failed once stays down: no more 10 s waits on flags that will never come)
    if (threadIdx.x == 0) ok_s = (*mine.err == 0 && (e < 3 || peer_wait(W, &mine.xack[q], e - 2, 1))) ? 1 : 0;       // the slot of epoch e - 2 is free again
    __syncthreads();
    if (!ok_s) return;
    double *dst = his.xbox + (static_cast<int64_t>(e & 1) * W.nranks + r) * W.cap_x;
    const double *src = send + X.off[k];
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // the last slice to land raises the flag (every block's stores are fenced before its arrival)
        if (__hip_atomic_fetch_add(&arrive[k], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
            __threadfence_system();
            __hip_atomic_store(&his.xflag[r], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ok_s = peer_wait(W, &mine.xflag[q], e, 1) ? 1 : 0;
    }
    __syncthreads();
    if (!ok_s) return;
    __threadfence_system();
    const double *box = mine.xbox + (static_cast<int64_t>(e & 1) * W.nranks + q) * W.cap_x;
    double *out = recv + X.off[k];
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) out[i] = __builtin_nontemporal_load(box + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        last_s = __hip_atomic_fetch_add(&arrive[kPeerMaxRanks + k], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1 ? 1 : 0;
        if (last_s) {
            __hip_atomic_store(&his.xack[r], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            W.xepoch[q] = e;
            __hip_atomic_store(&arrive[k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&arrive[kPeerMaxRanks + k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// in-place sum of n <= cap_a doubles over the ranks: every rank pushes its vector into every rank's box (its own too), then
// sums the boxes in rank order -- the same additions in the same order everywhere: identical bits on all ranks.
// Phase 1 (nranks x pb blocks: pb slices per destination -- one block per destination took 264 us for the 125 000-double
// right-hand side of a replicated multigrid level): push a slice; the last slice of a destination to land raises its flag.
// Phase 2 (same launch, after all flags): grid-stride sum.  The all-reduce number comes from the device (graph replay).
__global__ void __launch_bounds__(1024) k_peer_allreduce(PeerWorld W, double *__restrict__ d, int64_t n, int pb, int *arrive /* [1 + kPeerMaxRanks] */)
{
    __shared__ int ok_s;
    const int r = W.rank, nr = W.nranks;
    const unsigned long long e = *W.aepoch + 1;          // (advanced by k_peer_allreduce_ack, the launch behind this one)
    const PeerMail &mine = W.m[r];
    const bool pusher = static_cast<int>(blockIdx.x) < nr * pb;
    if (pusher) {
        const int q = blockIdx.x / pb, b = blockIdx.x % pb;
        const int64_t per = (n + pb - 1) / pb, i0 = per * b, i1 = i0 + per < n ? i0 + per : n;
        const PeerMail &his = W.m[q];
        if (threadIdx.x == 0) ok_s = (*mine.err == 0 && (e < 3 || peer_wait(W, &mine.aack[q], e - 2, 1))) ? 1 : 0;
        __syncthreads();
        if (ok_s) {
            double *dst = his.abox + (static_cast<int64_t>(e & 1) * nr + r) * W.cap_a;
            for (int64_t i = i0 + threadIdx.x; i < i1; i += 1024) dst[i] = d[i];
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0 && __hip_atomic_fetch_add(&arrive[1 + q], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == pb - 1) {
                __threadfence_system();
                __hip_atomic_store(&his.aflag[r], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // every block waits for all contributions (and for this rank's own pushes to have read d: the block-level arrival count)
    if (threadIdx.x == 0) {
        int ok = *mine.err == 0 ? 1 : 0;
        if (pusher) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (int q = 0; q < nr && ok; ++q) ok = peer_wait(W, &mine.aflag[q], e, 1) ? 1 : 0;
        if (ok) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nr * pb) {      // all pushes of this rank have read d
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > kPeerSpinTicks) { peer_fail(W); ok = 0; break; }
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    if (!ok_s) return;
    __threadfence_system();
    const double *box = mine.abox + static_cast<int64_t>(e & 1) * nr * W.cap_a;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 1024 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 1024) {
        double a = 0.0;
        for (int q = 0; q < nr; ++q) a += __builtin_nontemporal_load(box + q * W.cap_a + i);
        d[i] = a;
    }
}
// after the sum: tell every rank its box of this epoch has been read (separate launch: all blocks of the sum are done)
__global__ void k_peer_allreduce_ack(PeerWorld W, int *arrive)
{
    const int q = threadIdx.x;
    const unsigned long long e = *W.aepoch + 1;
    __syncthreads();
    if (q == 0) { *arrive = 0; *W.aepoch = e; }
    if (q < W.nranks) arrive[1 + q] = 0;
    if (q < W.nranks) __hip_atomic_store(&W.m[q].aack[W.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace pfem
