// Host side of the C ABI under AddressSanitizer + UBSan (CPU build only: GPU sanitizers are not
// available on this pool).  Exercises every host entry point of pfem_host.cpp on small inputs,
// including ragged / empty / out-of-contract ones.  Built and run by tests/test_sanitize.py.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/pfem_amd.h"
#include "../../pfemfort_amd/csrc/pfem_elem.hpp"

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    // --- structured mesh, slabs, numbering -----------------------------------------------
    const int nEx = 5, nEy = 4, nEz = 7, ndof = 3;
    const int64_t nNode = (nEx + 1) * (nEy + 1) * (nEz + 1), nElem = 6LL * nEx * nEy * nEz;
    int64_t nDBC = 0;
    assert(pfem_gen_box_tets(-1, 1, nEx, 0, 2, nEy, -1, 3, nEz, 0, nEz, 1, ndof, nullptr, nullptr, &nDBC, nullptr, nullptr, nullptr) == 0);
    std::vector<double> xyz(3 * nNode), bv(nDBC);
    std::vector<int32_t> conn(4 * nElem), bn(nDBC), bd(nDBC);
    assert(pfem_gen_box_tets(-1, 1, nEx, 0, 2, nEy, -1, 3, nEz, 0, nEz, 1, ndof, xyz.data(), conn.data(), &nDBC, bn.data(), bd.data(), bv.data()) == 0);
    std::vector<int32_t> slab(4 * 6 * nEx * nEy * 2);
    int64_t n2 = 0;
    assert(pfem_gen_box_tets(-1, 1, nEx, 0, 2, nEy, -1, 3, nEz, 2, 4, 0, 1, nullptr, slab.data(), &n2, nullptr, nullptr, nullptr) == 0);
    assert(pfem_gen_box_tets(-1, 1, 0, 0, 2, nEy, -1, 3, nEz, 0, nEz, 0, 1, nullptr, nullptr, &n2, nullptr, nullptr, nullptr) == PFEM_ERR_ARG);
    for (int parts : {1, 2, 3, 7}) {
        std::vector<int32_t> epid(nElem), npid(nNode), old_(nNode), new_(nNode), nda(nNode * ndof), edof(12 * nElem), assy(nNode * ndof);
        std::vector<double> sa(nNode * ndof);
        std::vector<int64_t> ns(parts), ne(parts), rs(parts), re(parts);
        int64_t N = 0;
        assert(pfem_partition_box_slabs(nEx, nEy, nEz, parts, epid.data(), npid.data()) == 0);
        assert(pfem_dof_numbering(nNode, ndof, nDBC, bn.data(), bd.data(), bv.data(), parts, npid.data(), old_.data(), new_.data(),
                                  nda.data(), sa.data(), ns.data(), ne.data(), rs.data(), re.data(), &N) == 0);
        assert(re[parts - 1] == N && rs[0] == 0);
        std::vector<int32_t> cn(4 * nElem);
        std::vector<double> xn(3 * nNode);
        assert(pfem_renumber_mesh(nNode, 3, nElem, 4, conn.data(), xyz.data(), new_.data(), old_.data(), cn.data(), xn.data()) == 0);
        for (size_t i = 0; i < cn.size(); ++i) assert(cn[i] == new_[conn[i]]);
        for (int64_t i = 0; i < nNode; ++i) assert(xn[i] == xyz[old_[i]]);
        assert(pfem_elem_dof_array(nElem, 4, ndof, cn.data(), nda.data(), edof.data()) == 0);
        assert(pfem_assy_for_soln(nNode, ndof, nda.data(), assy.data()) == 0);
        int64_t ng = 0;
        assert(pfem_find_ghosts(12 * nElem, edof.data(), rs[0], re[0] - rs[0], &ng, nullptr) == 0);
        std::vector<int64_t> gh(ng + 1);
        assert(pfem_find_ghosts(12 * nElem, edof.data(), rs[0], re[0] - rs[0], &ng, gh.data()) == 0);
        assert((parts == 1) == (ng == 0));
        // closed-form slab sizes against the bookkeeping above (z-slabs: the renumbering is the identity)
        for (int part = 0; part < parts; ++part) {
            int64_t sg = 0, r0 = 0, sl = 0, nn = 0, nel = 0;
            assert(pfem_box_slab_sizes(nEx, nEy, nEz, 1, ndof, parts, part, &sg, &r0, &sl, &nn, &nel) == 0);
            assert(sg == N && r0 == rs[part] && r0 + sl == re[part]);
            int64_t cnt = 0;
            for (int64_t e = 0; e < nElem; ++e) cnt += epid[e] == part;
            assert(nel == cnt);
        }
        // neighbour plan of every rank from everybody's ghost lists: symmetric, ascending, consistent with ownership
        std::vector<std::vector<int64_t>> ghosts(parts);
        std::vector<int64_t> goff(parts + 1, 0), gall;
        for (int r = 0; r < parts; ++r) {
            std::vector<int32_t> mine;
            for (int64_t e = 0; e < nElem; ++e)
                if (epid[e] == r) for (int i = 0; i < 12; ++i) mine.push_back(edof[i * nElem + e]);
            int64_t k = 0;
            assert(pfem_find_ghosts(static_cast<int64_t>(mine.size()), mine.data(), rs[r], re[r] - rs[r], &k, nullptr) == 0);
            ghosts[r].resize(k + 1);
            assert(pfem_find_ghosts(static_cast<int64_t>(mine.size()), mine.data(), rs[r], re[r] - rs[r], &k, ghosts[r].data()) == 0);
            ghosts[r].resize(k);
            goff[r + 1] = goff[r] + k;
            gall.insert(gall.end(), ghosts[r].begin(), ghosts[r].end());
        }
        gall.push_back(0);
        std::vector<std::vector<int>> peers(parts);
        std::vector<std::vector<int64_t>> poff(parts), pgid(parts);
        for (int r = 0; r < parts; ++r) {
            int np = 0; int64_t tot = 0;
            assert(pfem_neighbour_plan(parts, r, rs.data(), re.data(), goff.data(), gall.data(), &np, &tot, nullptr, nullptr, nullptr) == 0);
            peers[r].resize(np + 1); poff[r].assign(np + 1, 0); pgid[r].resize(tot + 1);
            if (np) assert(pfem_neighbour_plan(parts, r, rs.data(), re.data(), goff.data(), gall.data(), &np, &tot, peers[r].data(), poff[r].data(), pgid[r].data()) == 0);
            peers[r].resize(np);
            assert((parts == 1) == (np == 0));
        }
        for (int r = 0; r < parts; ++r)
            for (size_t k = 0; k < peers[r].size(); ++k) {
                const int q = peers[r][k];
                size_t kq = 0;
                while (kq < peers[q].size() && peers[q][kq] != r) ++kq;
                assert(kq < peers[q].size());                                        // q names r too ...
                const int64_t n = poff[r][k + 1] - poff[r][k];
                assert(n == poff[q][kq + 1] - poff[q][kq]);                           // ... with the same list
                assert(std::memcmp(&pgid[r][poff[r][k]], &pgid[q][poff[q][kq]], sizeof(int64_t) * n) == 0);
            }
    }
    {   // malformed plans are refused
        int64_t rs2[2] = {0, 4}, re2[2] = {4, 8}, off[3] = {0, 2, 2}, g[2] = {6, 5};
        int np; int64_t tot;
        assert(pfem_neighbour_plan(2, 0, rs2, re2, off, g, &np, &tot, nullptr, nullptr, nullptr) == PFEM_ERR_ARG);   // not ascending
        assert(pfem_neighbour_plan(2, 2, rs2, re2, off, g, &np, &tot, nullptr, nullptr, nullptr) == PFEM_ERR_ARG);
        assert(pfem_box_slab_sizes(4, 4, 4, 0, 1, 5, 0, nullptr, nullptr, nullptr, nullptr, nullptr) == PFEM_ERR_ARG);
    }
    assert(pfem_partition_box_slabs(nEx, nEy, nEz, nEz + 1, nullptr, nullptr) == PFEM_ERR_ARG);
    {   // out-of-range Dirichlet record is rejected, not written
        int32_t badn = static_cast<int32_t>(nNode), badd = 0; double v = 1;
        std::vector<int32_t> o(nNode), w(nNode), nda(nNode); std::vector<double> sa(nNode); int64_t a, b, c, d, N;
        assert(pfem_dof_numbering(nNode, 1, 1, &badn, &badd, &v, 1, nullptr, o.data(), w.data(), nda.data(), sa.data(), &a, &b, &c, &d, &N) == PFEM_ERR_ARG);
    }
    // --- per-element routines ---------------------------------------------------------------
    std::mt19937 rng(7);
    std::normal_distribution<double> nd(0, 1);
    for (int rep = 0; rep < 200; ++rep) {
        double x[4], y[4], z[4], K[144], F[12], vc[12] = {0}, ed[6] = {240.5, 0.3, 1.0, 0.1, 0.2, 0.3}, td[3] = {0, 1, 0};
        for (int i = 0; i < 4; ++i) { x[i] = nd(rng); y[i] = nd(rng); z[i] = nd(rng); }
        int a = pfem_poisson_tet_ke(x, y, z, ed, td, vc, K, F), b = pfem_elast_tet_ke(x, y, z, ed, td, vc, K, F);
        assert((a == 0 || a == PFEM_ERR_NEG_JAC) && a == b);
        {   // the per-node slice used by the gather assembly reproduces the full routine bit for bit
            double vr[4] = {nd(rng), nd(rng), nd(rng), nd(rng)}, Kp[16], Fp[4];
            const bool ok = pfem::poisson_tet(x, y, z, 1.5, 0.5, 2.0, 0.75, vr, Kp, Fp);
            for (int an = 0; an < 4; ++an) {
                double kc[4], kr[4], fa = 0.0;
                assert(pfem::poisson_tet_node(x, y, z, 1.5, 0.5, 2.0, 0.75, vr, an, true, kc, kr, fa) == ok);
                if (!ok) continue;
                assert(std::memcmp(&fa, &Fp[an], 8) == 0);
                for (int j = 0; j < 4; ++j) {
                    assert(std::memcmp(&kc[j], &Kp[j + 4 * an], 8) == 0);
                    assert(std::memcmp(&kr[j], &Kp[an + 4 * j], 8) == 0);
                }
            }
        }
        int c = pfem_poisson_tria_ke(x, y, ed, td, vc, K, F), d = pfem_elast_tria_ke(x, y, ed, td, vc, K, F);
        assert((c == 0 || c == PFEM_ERR_NEG_JAC) && c == d);
    }
    {   // The lean geometry / node routine of the gather kernels against the literal ones: every nonzero result has the
        // same bits, a zero may differ in sign only.  Random elements, structured-box elements (many exactly zero
        // differences) and coordinates that are -0.0 (what "-0.00000000" in a node file parses to).
        auto same = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a == 0.0 && b == 0.0); };
        std::mt19937 r2(11);
        std::normal_distribution<double> g2(0, 1);
        const double grid[5] = {-1.0, -0.5, -0.0, 0.0, 0.5};
        long zeros_seen = 0;
        for (int rep = 0; rep < 20000; ++rep) {
            double x[4], y[4], z[4];
            for (int i = 0; i < 4; ++i) {
                if (rep & 1) { x[i] = g2(r2); y[i] = g2(r2); z[i] = g2(r2); }
                else { x[i] = grid[r2() % 5]; y[i] = grid[r2() % 5]; z[i] = grid[r2() % 5]; }
            }
            pfem::TetGeom a, b;
            pfem::tet_geometry(x, y, z, a);
            pfem::tet_geometry_lean(x, y, z, b);
            if (!(a.jac != 0.0) || !std::isfinite(1.0 / a.jac)) continue;      // degenerate: neither form is meaningful
            assert(same(a.jac, b.jac));
            for (int i = 0; i < 4; ++i) {
                assert(same(a.gx[i], b.gx[i]) && same(a.gy[i], b.gy[i]) && same(a.gz[i], b.gz[i]));
                zeros_seen += (a.gx[i] == 0.0) + (a.gy[i] == 0.0) + (a.gz[i] == 0.0);
            }
            const double vz[4] = {0.0, 0.0, 0.0, 0.0};
            for (int an = 0; an < 4; ++an) {
                double kc[4], kr[4], fa = 0.0, kc2[4], kr2[4], fa2 = 0.0;
                const bool ok = pfem::poisson_tet_node(x, y, z, 1.5, 0.5, 2.0, 0.75, vz, an, true, kc, kr, fa);
                assert(pfem::poisson_tet_node_lean(x, y, z, 1.5, 0.5, 2.0, 0.75, an, true, kc2, kr2, fa2) == ok);
                if (!ok) continue;
                assert(same(fa, fa2));
                for (int j = 0; j < 4; ++j) assert(same(kc[j], kc2[j]) && same(kr[j], kr2[j]));
                // ... and what reaches the matrix: an accumulator that starts at +0.0 ends with the same bits
                for (int j = 0; j < 4; ++j) {
                    double acc1 = 0.0, acc2 = 0.0;
                    acc1 += kc[j]; acc2 += kc2[j];
                    assert(std::memcmp(&acc1, &acc2, 8) == 0);
                }
            }
        }
        assert(zeros_seen > 1000);          // the zero-sign cases were really exercised
    }
    assert(pfem_poisson_tet_ke(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == PFEM_ERR_ARG);
    // --- ASCII ingest: ragged / empty / unterminated / huge tokens ------------------------------
    for (const char *txt : {"", "\n\n", "1 2 3", "1 2 3\n4 5 6\n", " 1\t2  3 \r\n4 5 6 7 8\n", "1 2 3\n4 5\n", "x y z\n",
                            "1 2 9999999999999999999999999999999999999999999999999999999999999999999999999999\n"}) {
        int64_t rows = -1; int cols = -1;
        int rc = pfem_text_table_shape(txt, static_cast<int64_t>(std::strlen(txt)), &rows, &cols);
        if (rc == 0 && rows > 0) {
            std::vector<double> out(rows * cols);
            (void)pfem_text_table_parse(txt, static_cast<int64_t>(std::strlen(txt)), rows, cols, out.data());
        }
    }
    // --- VTK writer ------------------------------------------------------------------------------
    {
        std::vector<int32_t> pid(nElem, 1);
        std::vector<double> sol(nNode * 3, -0.0);
        const std::string p = tmp + "/san.vtk";
        assert(pfem_write_vtk(p.c_str(), 3, nElem, nNode, 4, 3, xyz.data(), conn.data(), pid.data(), sol.data()) == 0);
        assert(pfem_write_vtk("/nonexistent-dir/x.vtk", 3, nElem, nNode, 4, 1, xyz.data(), conn.data(), pid.data(), sol.data()) == PFEM_ERR_ARG);
        std::remove(p.c_str());
        // wide and non-finite values take the snprintf path of the F12.6 formatter
        std::vector<double> odd(nNode, 0.0);
        for (int64_t i = 0; i < nNode; ++i) odd[i] = (i % 5 == 0) ? 1e300 : ((i % 5 == 1) ? -123456.789 : ((i % 5 == 2) ? std::nan("") : 0.4999995));
        assert(pfem_write_vtk(p.c_str(), 3, nElem, nNode, 4, 1, xyz.data(), conn.data(), pid.data(), odd.data()) == 0);
        std::remove(p.c_str());
        // temp.dat: both record forms, empty input, argument errors
        std::vector<int64_t> ii(nNode), ind(nNode);
        for (int64_t i = 0; i < nNode; ++i) { ii[i] = i + 1; ind[i] = nNode - i; }
        const std::string t = tmp + "/temp.dat";
        assert(pfem_write_temp_dat(t.c_str(), nNode, ii.data(), ind.data(), odd.data()) == 0);
        assert(pfem_write_temp_dat(t.c_str(), nNode, nullptr, nullptr, odd.data()) == 0);
        assert(pfem_write_temp_dat(t.c_str(), 0, nullptr, nullptr, nullptr) == 0);
        assert(pfem_write_temp_dat(t.c_str(), nNode, ii.data(), nullptr, odd.data()) == PFEM_ERR_ARG);
        assert(pfem_write_temp_dat("/nonexistent-dir/t.dat", 1, nullptr, nullptr, odd.data()) == PFEM_ERR_ARG);
        std::remove(t.c_str());
    }
    {   // recursive coordinate bisection: more parts than elements, one part, 2-D input, bad connectivity
        std::vector<int32_t> ep(nElem), np(nNode);
        for (int parts : {1, 2, 3, 7, 64})
            assert(pfem_partition_rcb(nNode, 3, xyz.data(), nElem, 4, conn.data(), parts, ep.data(), np.data()) == 0);
        assert(pfem_partition_rcb(nNode, 2, xyz.data(), nElem, 4, conn.data(), 3, ep.data(), np.data()) == 0);
        std::vector<int32_t> bad(conn);
        bad[5] = static_cast<int32_t>(nNode);
        assert(pfem_partition_rcb(nNode, 3, xyz.data(), nElem, 4, bad.data(), 3, ep.data(), np.data()) == PFEM_ERR_ARG);
        assert(pfem_partition_rcb(nNode, 3, xyz.data(), 0, 4, nullptr, 3, ep.data(), np.data()) == 0);
        assert(pfem_partition_rcb(nNode, 4, xyz.data(), nElem, 4, conn.data(), 3, ep.data(), np.data()) == PFEM_ERR_ARG);
    }
    std::puts("host_sanitize: ok");
    return 0;
}
