// ilu0_probe.hip -- LAB MEASUREMENT, not part of the product: what one application of the reference's preconditioner
// (PCBJACOBI -> ILU(0) per rank, solverpetsc.F:187,206) costs on MI355X with the vendor's production triangular solve.
// Reads a CSR matrix (int32 rowptr/cols, f64 vals; the assembled matrix of a BASELINE configuration, dumped by
// tools/probe_ilu0_gpu.py), runs rocsparse_dcsrilu0 in place, analyses L (unit lower) and U (upper) and times
// rocsparse_dcsrsv_solve for both.  Prints one JSON line.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/ilu0_probe.hip -lrocsparse -o tools/lab/ilu0_probe
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { std::fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); return 1; } } while (0)

template <class T>
static std::vector<T> slurp(const char *path)
{
    std::vector<T> v;
    FILE *f = std::fopen(path, "rb");
    if (!f) return v;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize(static_cast<size_t>(n) / sizeof(T));
    if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) v.clear();
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: ilu0_probe rowptr.i32 cols.i32 vals.f64 [reps]\n"); return 2; }
    const int reps = argc > 4 ? std::atoi(argv[4]) : 10;
    auto rowptr = slurp<int>(argv[1]);
    auto cols = slurp<int>(argv[2]);
    auto vals = slurp<double>(argv[3]);
    if (rowptr.size() < 2 || cols.size() != vals.size()) { std::fprintf(stderr, "bad input files\n"); return 2; }
    const int m = static_cast<int>(rowptr.size()) - 1, nnz = static_cast<int>(cols.size());

    int *d_ptr, *d_col;
    double *d_val, *d_x, *d_y, *d_z;
    CK(hipMalloc(&d_ptr, sizeof(int) * (m + 1)));
    CK(hipMalloc(&d_col, sizeof(int) * nnz));
    CK(hipMalloc(&d_val, sizeof(double) * nnz));
    CK(hipMalloc(&d_x, sizeof(double) * m));
    CK(hipMalloc(&d_y, sizeof(double) * m));
    CK(hipMalloc(&d_z, sizeof(double) * m));
    CK(hipMemcpy(d_ptr, rowptr.data(), sizeof(int) * (m + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, cols.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, vals.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    std::vector<double> ones(m, 1.0);
    CK(hipMemcpy(d_x, ones.data(), sizeof(double) * m, hipMemcpyHostToDevice));

    rocsparse_handle h;
    CK(rocsparse_create_handle(&h));
    rocsparse_mat_descr dM, dL, dU;
    CK(rocsparse_create_mat_descr(&dM));
    CK(rocsparse_create_mat_descr(&dL));
    CK(rocsparse_set_mat_fill_mode(dL, rocsparse_fill_mode_lower));
    CK(rocsparse_set_mat_diag_type(dL, rocsparse_diag_type_unit));
    CK(rocsparse_create_mat_descr(&dU));
    CK(rocsparse_set_mat_fill_mode(dU, rocsparse_fill_mode_upper));
    CK(rocsparse_set_mat_diag_type(dU, rocsparse_diag_type_non_unit));
    rocsparse_mat_info info;
    CK(rocsparse_create_mat_info(&info));
    size_t b0 = 0, b1 = 0, b2 = 0;
    CK(rocsparse_dcsrilu0_buffer_size(h, m, nnz, dM, d_val, d_ptr, d_col, info, &b0));
    CK(rocsparse_dcsrsv_buffer_size(h, rocsparse_operation_none, m, nnz, dL, d_val, d_ptr, d_col, info, &b1));
    CK(rocsparse_dcsrsv_buffer_size(h, rocsparse_operation_none, m, nnz, dU, d_val, d_ptr, d_col, info, &b2));
    size_t bs = b0 > b1 ? b0 : b1;
    if (b2 > bs) bs = b2;
    void *buf;
    CK(hipMalloc(&buf, bs));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_fact_an = 0, ms_fact = 0, ms_an = 0, ms_L = 0, ms_U = 0;
    CK(hipEventRecord(e0));
    CK(rocsparse_dcsrilu0_analysis(h, m, nnz, dM, d_val, d_ptr, d_col, info, rocsparse_analysis_policy_reuse, rocsparse_solve_policy_auto, buf));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_fact_an, e0, e1));
    CK(hipEventRecord(e0));
    CK(rocsparse_dcsrilu0(h, m, nnz, dM, d_val, d_ptr, d_col, info, rocsparse_solve_policy_auto, buf));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_fact, e0, e1));
    rocsparse_int pivot = -1;
    const rocsparse_status zp = rocsparse_csrilu0_zero_pivot(h, info, &pivot);
    CK(hipEventRecord(e0));
    CK(rocsparse_dcsrsv_analysis(h, rocsparse_operation_none, m, nnz, dL, d_val, d_ptr, d_col, info, rocsparse_analysis_policy_reuse, rocsparse_solve_policy_auto, buf));
    CK(rocsparse_dcsrsv_analysis(h, rocsparse_operation_none, m, nnz, dU, d_val, d_ptr, d_col, info, rocsparse_analysis_policy_reuse, rocsparse_solve_policy_auto, buf));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_an, e0, e1));
    const double one = 1.0;
    for (int w = 0; w < 2; ++w) {     // warm-up
        CK(rocsparse_dcsrsv_solve(h, rocsparse_operation_none, m, nnz, &one, dL, d_val, d_ptr, d_col, info, d_x, d_y, rocsparse_solve_policy_auto, buf));
        CK(rocsparse_dcsrsv_solve(h, rocsparse_operation_none, m, nnz, &one, dU, d_val, d_ptr, d_col, info, d_y, d_z, rocsparse_solve_policy_auto, buf));
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        CK(rocsparse_dcsrsv_solve(h, rocsparse_operation_none, m, nnz, &one, dL, d_val, d_ptr, d_col, info, d_x, d_y, rocsparse_solve_policy_auto, buf));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_L, e0, e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        CK(rocsparse_dcsrsv_solve(h, rocsparse_operation_none, m, nnz, &one, dU, d_val, d_ptr, d_col, info, d_y, d_z, rocsparse_solve_policy_auto, buf));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_U, e0, e1));
    std::vector<double> z(m);
    CK(hipMemcpy(z.data(), d_z, sizeof(double) * m, hipMemcpyDeviceToHost));
    double zs = 0;
    for (double v : z) zs += v;
    std::printf("{\"rows\": %d, \"nnz\": %d, \"ilu0_analysis_ms\": %.3f, \"ilu0_factorise_ms\": %.3f, \"csrsv_analysis_ms\": %.3f, "
                "\"lower_solve_ms\": %.4f, \"upper_solve_ms\": %.4f, \"reps\": %d, \"zero_pivot\": %d, \"checksum\": %.10e, "
                "\"library\": \"rocSPARSE csrilu0 + csrsv (ROCm 7.2), solve policy auto\"}\n",
                m, nnz, ms_fact_an, ms_fact, ms_an, ms_L / reps, ms_U / reps, reps, zp == rocsparse_status_zero_pivot ? (int)pivot : -1, zs);
    return 0;
}
