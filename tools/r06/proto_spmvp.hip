// Prototype (round 6 lab): a pattern-structured SpMV -- lane = row, column = row + offset[pattern][k], values as 16-bit codes
// (16 per row, 32 B, row-major) into an LDS dictionary.  Synthetic data of config 3's shape (199^3 rows, 15-point Kuhn stencil,
// 601 values).  Question: can this form beat k_spmvr_vd's 102 us, and what bounds it?   hipcc --offload-arch=gfx950 -O3
//   variant A: every entry's x from memory (15 gathers of 8 B per lane)
//   variant C: an entry whose offset is the previous entry's + 1 shifts the previous entry's x by one lane (DPP); the wave's last
//              lane loads its own (15 memory instructions still, 8 of them for one lane)
//   variant E: the last lane's values of all shifted entries come from ONE load (lane e loads the e-th): 7 + 1 memory instructions
// PROTO_STENCIL=1: all offsets within the line; 3: all offsets 0 (perfect L1 hits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)
constexpr int kBlock = 256, kMaxPat = 64, kW = 16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int dpp_shl1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
template <int VAR, int SEG, bool WITH_DOT>
__global__ void __launch_bounds__(kBlock) k_spmvp(int64_t n, const u32x4 *__restrict__ codes, const uint16_t *__restrict__ pid, const int *__restrict__ tbl,
                                                  const int *__restrict__ tlen, const unsigned *__restrict__ tflag, int npat, const double *__restrict__ dict, int nd,
                                                  const double *__restrict__ x, double *__restrict__ y, double *__restrict__ partial)
{
    extern __shared__ double vd[];
    __shared__ int offs[kMaxPat * kW];
    __shared__ int lens[kMaxPat];
    __shared__ int eoffs[kMaxPat * 8];
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t R = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * (64 * SEG);
    int p[SEG];
    u32x4 q0[SEG], q1[SEG];
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        const int64_t row = R + 64 * j + lane;
        const bool live = row < n;
        p[j] = live ? pid[row] : 0;
        q0[j] = live ? __builtin_nontemporal_load(codes + 2 * row) : u32x4{0, 0, 0, 0};
        q1[j] = live ? __builtin_nontemporal_load(codes + 2 * row + 1) : u32x4{0, 0, 0, 0};
    }
    for (int i = threadIdx.x; i < nd; i += kBlock) vd[i] = dict[i];
    for (int i = threadIdx.x; i < npat * kW; i += kBlock) offs[i] = tbl[i];
    if (threadIdx.x < npat) lens[threadIdx.x] = tlen[threadIdx.x];
    if (VAR >= 2)
        for (int i = threadIdx.x; i < npat * 8; i += kBlock) {          // e-th shifted entry's offset of pattern i / 8
            const unsigned f = tflag[i >> 3];
            int e = i & 7, off = 0;
            for (int k = 0; k < 15; ++k)
                if ((f >> k) & 1u) { if (e == 0) off = tbl[(i >> 3) * kW + k]; --e; }
            eoffs[i] = off;
        }
    __syncthreads();
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        const int64_t row = R + 64 * j + lane;
        const bool live = row < n;
        const int p0 = __builtin_amdgcn_readfirstlane(p[j]);
        const uint32_t qw[8] = {q0[j].x, q0[j].y, q0[j].z, q0[j].w, q1[j].x, q1[j].y, q1[j].z, q1[j].w};
        double xv[15];
        if (VAR == 4) {
#pragma unroll
            for (int k = 0; k < 15; ++k) xv[k] = 1.0 + 0.5 * k;          // no x at all: what the code stream + dictionary + fma chain cost
        } else if (VAR == 5) {
            const double x0 = live ? x[row] : 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) xv[k] = x0 + 0.5 * k;           // one x per row
        } else if (__all(p[j] == p0 && live) && (VAR != 3 || tflag[p0] == 0x55aau)) {
            const int len = tlen[p0];
            const unsigned fl = VAR == 0 ? 0u : (VAR == 3 ? 0x55aau : tflag[p0]);
            const int *__restrict__ t = tbl + p0 * kW;
            if (VAR >= 2) {
                const int nsh = __builtin_popcount(fl);
                const double ev = lane < nsh ? x[R + 64 * j + 63 + eoffs[p0 * 8 + lane]] : 0.0;
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    xv[k] = 0.0;
                    if (!((fl >> k) & 1u) && k < len) xv[k] = x[row + t[k]];
                }
                const int elo = __double2loint(ev), ehi = __double2hiint(ev);
#pragma unroll
                for (int k = 1; k < 15; ++k)
                    if ((fl >> k) & 1u) {
                        const int e = __builtin_popcount(fl & ((1u << k) - 1u));
                        const int slo = dpp_shl1(__double2loint(xv[k - 1])), shi = dpp_shl1(__double2hiint(xv[k - 1]));
                        const int blo = __builtin_amdgcn_readlane(elo, e), bhi = __builtin_amdgcn_readlane(ehi, e);
                        xv[k] = __hiloint2double(lane == 63 ? bhi : shi, lane == 63 ? blo : slo);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    const bool sh = (fl >> k) & 1u;
                    xv[k] = (k < len && (!sh || lane == 63)) ? x[row + t[k]] : 0.0;
                }
#pragma unroll
                for (int k = 1; k < 15; ++k)
                    if ((fl >> k) & 1u) {
                        const int slo = dpp_shl1(__double2loint(xv[k - 1])), shi = dpp_shl1(__double2hiint(xv[k - 1]));
                        xv[k] = lane == 63 ? xv[k] : __hiloint2double(shi, slo);
                    }
            }
        } else {
            const int len = lens[p[j]];
#pragma unroll
            for (int k = 0; k < 15; ++k) xv[k] = (live && k < len) ? x[row + offs[p[j] * kW + k]] : 0.0;
        }
        double acc = vd[qw[0] & 0xffffu] * xv[0];
#pragma unroll
        for (int k = 1; k < 15; ++k) {
            const uint32_t c = (k & 1) ? (qw[k >> 1] >> 16) : (qw[k >> 1] & 0xffffu);
            acc = __builtin_fma(vd[c], xv[k], acc);
        }
        if (live) {
            __builtin_nontemporal_store(acc, y + row);
            if (WITH_DOT) dot = __builtin_fma(x[row], acc, dot);
        }
    }
    if (WITH_DOT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        if (lane == 0) sm[wave] = dot;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}
struct Args { int64_t n; const u32x4 *codes; const uint16_t *pid; const int *tbl, *tlen; const unsigned *tflag; int npat; const double *dict; int nd; const double *x; double *y, *partial; };
template <int VAR, int SEG, bool WITH_DOT>
void run(const Args &A, const char *name, double bytes)
{
    const unsigned grid = static_cast<unsigned>((A.n + 256 * SEG - 1) / (256 * SEG));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&] { hipLaunchKernelGGL((k_spmvp<VAR, SEG, WITH_DOT>), dim3(grid), dim3(kBlock), A.nd * 8, 0, A.n, A.codes, A.pid, A.tbl, A.tlen, A.tflag, A.npat, A.dict, A.nd, A.x, A.y, A.partial); };
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms / 50 * 1000.0;
    std::printf("%-28s %6.1f us  %.2f TB/s of the minimal bytes\n", name, t, bytes / t / 1e6);
}
int main()
{
    const int m = 199;
    const int64_t n = static_cast<int64_t>(m) * m * m;
    const int nd = 601, npat = 64;
    const int64_t a = m, b = static_cast<int64_t>(m) * m;
    std::vector<int64_t> st = {-(1 + a + b), -(a + b), -(1 + b), -b, -(1 + a), -a, -1, 0, 1, a, 1 + a, b, 1 + b, a + b, 1 + a + b};
    const int mode = std::getenv("PROTO_STENCIL") ? std::atoi(std::getenv("PROTO_STENCIL")) : 0;
    if (mode == 1) st = {-7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7};
    if (mode == 3) st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int> tbl(kMaxPat * kW, 0), tlen(kMaxPat, 0);
    std::vector<uint16_t> pid(n);
    if (mode == 0) {
        // pattern p = 9 fz + 3 fy + fx, f* in {0: at the low face, 1: interior, 2: at the high face}: entries that leave the box are dropped
        for (int p = 0; p < 27; ++p) {
            const int fx = p % 3, fy = (p / 3) % 3, fz = p / 9;
            int len = 0;
            for (int64_t o : st) {
                int64_t r = o;
                int dz = 0, dy = 0;
                if (r > b / 2) { dz = 1; r -= b; } else if (r < -b / 2) { dz = -1; r += b; }
                if (r > a / 2) { dy = 1; r -= a; } else if (r < -a / 2) { dy = -1; r += a; }
                const int dx = static_cast<int>(r);
                if ((fx == 0 && dx < 0) || (fx == 2 && dx > 0) || (fy == 0 && dy < 0) || (fy == 2 && dy > 0) || (fz == 0 && dz < 0) || (fz == 2 && dz > 0)) continue;
                tbl[p * kW + len++] = static_cast<int>(o);
            }
            tlen[p] = len;
        }
        for (int64_t r = 0; r < n; ++r) {
            const int ix = r % m, iy = (r / m) % m, iz = r / b;
            pid[r] = static_cast<uint16_t>((ix == 0 ? 0 : ix == m - 1 ? 2 : 1) + 3 * (iy == 0 ? 0 : iy == m - 1 ? 2 : 1) + 9 * (iz == 0 ? 0 : iz == m - 1 ? 2 : 1));
        }
    } else {
        for (int k = 0; k < 15; ++k) tbl[13 * kW + k] = static_cast<int>(st[k]);
        tlen[13] = 15; tbl[63 * kW] = 0; tlen[63] = 1;
        for (int64_t r = 0; r < n; ++r) pid[r] = (r < 3 * b || r >= n - 3 * b) ? 63 : 13;
    }
    std::vector<unsigned> tflag(kMaxPat, 0);
    for (int q = 0; q < kMaxPat; ++q)
        for (int k = 1; k < tlen[q]; ++k)
            if (tbl[q * kW + k] == tbl[q * kW + k - 1] + 1) tflag[q] |= 1u << k;
    std::vector<uint16_t> codes(static_cast<size_t>(n) * 16);
    uint32_t s = 12345;
    // (values repeat along the lines, as the assembled matrix's do: the same code for an entry of neighbouring rows, a different one now and then)
    for (int64_t r = 0; r < n; ++r)
        for (int k = 0; k < 16; ++k) {
            s = s * 1664525u + 1013904223u;
            codes[static_cast<size_t>(r) * 16 + k] = static_cast<uint16_t>((k * 37 + ((r / 97) % 5) * 3 + ((s >> 8) % 64 == 0 ? 1 : 0)) % nd);
        }
    std::vector<double> dict(nd), x(n);
    for (int i = 0; i < nd; ++i) dict[i] = 0.001 * (i - 300);
    for (int64_t i = 0; i < n; ++i) x[i] = 1.0 + 1e-3 * (i % 977);
    u32x4 *d_codes; uint16_t *d_pid; int *d_tbl, *d_tlen; unsigned *d_tflag; double *d_dict, *d_x, *d_y, *d_part;
    CK(hipMalloc(&d_codes, codes.size() * 2)); CK(hipMalloc(&d_pid, n * 2)); CK(hipMalloc(&d_tbl, tbl.size() * 4)); CK(hipMalloc(&d_tlen, tlen.size() * 4));
    CK(hipMalloc(&d_tflag, kMaxPat * 4)); CK(hipMalloc(&d_dict, nd * 8)); CK(hipMalloc(&d_x, n * 8)); CK(hipMalloc(&d_y, n * 8)); CK(hipMalloc(&d_part, 1 << 20));
    CK(hipMemcpy(d_codes, codes.data(), codes.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pid, pid.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tlen, tlen.data(), tlen.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tflag, tflag.data(), kMaxPat * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dict, dict.data(), nd * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_x, x.data(), n * 8, hipMemcpyHostToDevice));
    const double bytes = n * (32.0 + 2.0 + 8.0 + 8.0);
    std::printf("rows %lld, minimal bytes %.1f MB, interior pattern: %d entries, shift flags %x\n", static_cast<long long>(n), bytes / 1e6, tlen[13], tflag[13]);
    const Args A{n, d_codes, d_pid, d_tbl, d_tlen, d_tflag, npat, d_dict, nd, d_x, d_y, d_part};
    run<0, 1, false>(A, "A (15 gathers) SEG 1", bytes);
    run<0, 2, false>(A, "A (15 gathers) SEG 2", bytes);
    run<0, 4, false>(A, "A (15 gathers) SEG 4", bytes);
    run<0, 2, true>(A, "A (15 gathers) SEG 2 + dot", bytes);
    run<1, 2, false>(A, "C (7 + 8 one-lane) SEG 2", bytes);
    run<1, 4, false>(A, "C (7 + 8 one-lane) SEG 4", bytes);
    run<2, 1, false>(A, "E (7 + 1) SEG 1", bytes);
    run<2, 2, false>(A, "E (7 + 1) SEG 2", bytes);
    run<2, 4, false>(A, "E (7 + 1) SEG 4", bytes);
    run<2, 2, true>(A, "E (7 + 1) SEG 2 + dot", bytes);
    run<4, 2, false>(A, "G (no x) SEG 2", bytes);
    run<4, 4, false>(A, "G (no x) SEG 4", bytes);
    run<5, 2, false>(A, "H (one x per row) SEG 2", bytes);
    run<5, 4, false>(A, "H (one x per row) SEG 4", bytes);
    return 0;
    run<3, 1, false>(A, "F (7 + 1, shape known) SEG 1", bytes);
    run<3, 2, false>(A, "F (7 + 1, shape known) SEG 2", bytes);
    run<3, 4, false>(A, "F (7 + 1, shape known) SEG 4", bytes);
    run<3, 2, true>(A, "F (7 + 1, shape known) SEG 2 + dot", bytes);
    std::vector<double> y(n);
    CK(hipMemcpy(y.data(), d_y, n * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int64_t r = 0; r < n; r += 997) {
        const int p = pid[r];
        double acc = 0;
        for (int k = 0; k < tlen[p]; ++k) acc = k == 0 ? dict[codes[r * 16 + k]] * x[r + tbl[p * kW + k]] : __builtin_fma(dict[codes[r * 16 + k]], x[r + tbl[p * kW + k]], acc);
        worst = std::max(worst, std::abs(acc - y[r]));
    }
    std::printf("max abs difference to the host on a sample (last variant run): %g\n", worst);
    return 0;
}
