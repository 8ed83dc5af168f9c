// pfem_elem.hpp -- element arithmetic shared by the device kernels and the host
// per-element C-ABI entry points.
//
// One source for both sides so the per-element compat surface (host) and the batched
// assembly kernels (gfx950) produce the same bits.  Evaluation ORDER follows the
// reference statement by statement and the translation unit is compiled with
// -ffp-contract=off, because the reference is built without FMA contraction; with
// IEEE division on both sides the element matrices are bit-identical to the
// reference's (tests/test_elements_*).
//
// Reference: elementutilitiesbasisfuncs.F:16-52,165-234 (tria), :242-289,430-538 (tet);
//            elementutilitiespoisson.F:23-101,107-193;
//            elementutilitieselasticity3D.F:248-393;
//            triapoissonserialimpl1.F:573-594.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PFEM_HD __host__ __device__ __forceinline__
#else
#define PFEM_HD inline
#endif

namespace pfem {

// REAL(4) literals widened to double (the reference is compiled without
// -fdefault-real-8): 1.0/6.0 -> 0.16666667163372040, 1.0/3.0 -> 0.3333333432674408
constexpr double kGaussWtTet = static_cast<double>(1.0f / 6.0f);
constexpr double kGaussPtTria = static_cast<double>(1.0f / 3.0f);

// ---------------------------------------------------------------------------
// P1 tetrahedron: physical gradients and Jacobian at the single Gauss point.
// Local node order (a,b,c,d); parametric gradients g_a=(1,0,0), g_b=(0,1,0),
// g_c=(-1,-1,-1), g_d=(0,0,1)   [elementutilitiesbasisfuncs.F:268-281].
// The mapping matrix rows are a-c, b-c, d-c [:493-509]: the reference
// accumulates 0 + x_a*1 + x_b*0 + x_c*(-1) + x_d*0, which equals x_a - x_c
// exactly in IEEE arithmetic for finite inputs.
// ---------------------------------------------------------------------------
struct TetGeom {
    double gx[4], gy[4], gz[4];  // dN_a/dx, dN_a/dy, dN_a/dz
    double jac;
};

PFEM_HD void tet_geometry(const double x[4], const double y[4], const double z[4], TetGeom &g)
{
    // zero-initialised accumulation reproduced literally for the signed-zero cases
    const double b11 = ((0.0 + x[0]) + x[1] * 0.0) + (x[2] * -1.0) + x[3] * 0.0;
    const double b21 = ((0.0 + x[0] * 0.0) + x[1]) + (x[2] * -1.0) + x[3] * 0.0;
    const double b31 = ((0.0 + x[0] * 0.0) + x[1] * 0.0) + (x[2] * -1.0) + x[3];
    const double b12 = ((0.0 + y[0]) + y[1] * 0.0) + (y[2] * -1.0) + y[3] * 0.0;
    const double b22 = ((0.0 + y[0] * 0.0) + y[1]) + (y[2] * -1.0) + y[3] * 0.0;
    const double b32 = ((0.0 + y[0] * 0.0) + y[1] * 0.0) + (y[2] * -1.0) + y[3];
    const double b13 = ((0.0 + z[0]) + z[1] * 0.0) + (z[2] * -1.0) + z[3] * 0.0;
    const double b23 = ((0.0 + z[0] * 0.0) + z[1]) + (z[2] * -1.0) + z[3] * 0.0;
    const double b33 = ((0.0 + z[0] * 0.0) + z[1] * 0.0) + (z[2] * -1.0) + z[3];

    // determinant, three-term cofactor expansion in the reference's order [:512-514]
    double jac = b11 * (b22 * b33 - b23 * b32);
    jac = jac + b12 * (b23 * b31 - b21 * b33);
    jac = jac + b13 * (b21 * b32 - b22 * b31);
    const double di = 1.0 / jac;  // [:517]

    // adjugate inverse [:520-528]; i(r,c) == Binv(r,c)
    const double i11 = +di * (b22 * b33 - b23 * b32);
    const double i21 = -di * (b21 * b33 - b23 * b31);
    const double i31 = +di * (b21 * b32 - b22 * b31);
    const double i12 = -di * (b12 * b33 - b13 * b32);
    const double i22 = +di * (b11 * b33 - b13 * b31);
    const double i32 = -di * (b11 * b32 - b12 * b31);
    const double i13 = +di * (b12 * b23 - b13 * b22);
    const double i23 = -di * (b11 * b23 - b13 * b21);
    const double i33 = +di * (b11 * b22 - b12 * b21);

    // dN/dx = g.(Binv row 1) etc. [:532-536], with the literal 1/0/-1 products kept
    const double u1[4] = {1.0, 0.0, -1.0, 0.0};
    const double u2[4] = {0.0, 1.0, -1.0, 0.0};
    const double u3[4] = {0.0, 0.0, -1.0, 1.0};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        g.gx[a] = u1[a] * i11 + u2[a] * i12 + u3[a] * i13;
        g.gy[a] = u1[a] * i21 + u2[a] * i22 + u3[a] * i23;
        g.gz[a] = u1[a] * i31 + u2[a] * i32 + u3[a] * i33;
    }
    g.jac = jac;
}

// The same geometry without the reference's literal products by 0.0 / 1.0 / -1.0 and its zero-initialised sums (117 of
// the 168 operations above).  For finite coordinates every NONZERO result has the same bits: x + (+-0) = x, x * -1.0 = -x,
// and the remaining additions are the same additions in the same order ((-x2) + x3 = x3 - x2 exactly).  What may differ is
// the SIGN OF A ZERO entry (the literal form turns -0 into +0 at its first `0.0 +`).  A signed zero can only travel on as
// a signed zero (no division by it short of a degenerate element, which the reference cannot assemble either), and every
// value the gather kernels build from these is ADDED to an accumulator that starts at +0.0, where +-0 changes nothing:
// assembled K and F have the same bits either way.  Used by the gather kernels only; the per-element entry points and
// pfem_eval_elems keep the literal form above.  (tests/native: lean == literal up to the sign of zeros, -0.0 coordinates
// included; tests/test_gpu_*: assembled K, F bit-exact against the oracle's literal serial loop.)
PFEM_HD void tet_geometry_lean(const double x[4], const double y[4], const double z[4], TetGeom &g)
{
    const double b11 = x[0] - x[2], b21 = x[1] - x[2], b31 = x[3] - x[2];
    const double b12 = y[0] - y[2], b22 = y[1] - y[2], b32 = y[3] - y[2];
    const double b13 = z[0] - z[2], b23 = z[1] - z[2], b33 = z[3] - z[2];
    double jac = b11 * (b22 * b33 - b23 * b32);
    jac = jac + b12 * (b23 * b31 - b21 * b33);
    jac = jac + b13 * (b21 * b32 - b22 * b31);
    const double di = 1.0 / jac;
    const double i11 = +di * (b22 * b33 - b23 * b32);
    const double i21 = -di * (b21 * b33 - b23 * b31);
    const double i31 = +di * (b21 * b32 - b22 * b31);
    const double i12 = -di * (b12 * b33 - b13 * b32);
    const double i22 = +di * (b11 * b33 - b13 * b31);
    const double i32 = -di * (b11 * b32 - b12 * b31);
    const double i13 = +di * (b12 * b23 - b13 * b22);
    const double i23 = -di * (b11 * b23 - b13 * b21);
    const double i33 = +di * (b11 * b22 - b12 * b21);
    g.gx[0] = i11;  g.gx[1] = i12;  g.gx[2] = (-i11 - i12) - i13;  g.gx[3] = i13;
    g.gy[0] = i21;  g.gy[1] = i22;  g.gy[2] = (-i21 - i22) - i23;  g.gy[3] = i23;
    g.gz[0] = i31;  g.gz[1] = i32;  g.gz[2] = (-i31 - i32) - i33;  g.gz[3] = i33;
    g.jac = jac;
}

// Poisson on a P1 tet, one Gauss point (1/4,1/4,1/4), source term -6
// [elementutilitiespoisson.F:107-193].  K column-major 4x4.  Returns false when the
// reference would STOP on a negative Jacobian [:157].
PFEM_HD bool poisson_tet(const double x[4], const double y[4], const double z[4], double kx,
                         double ky, double kz, double af, const double valC[4], double K[16],
                         double F[4])
{
    TetGeom g;
    tet_geometry(x, y, z, g);
    if (g.jac < 0.0) return false;
    const double dvol = kGaussWtTet * g.jac;                 // [:161]
    const double N[4] = {0.25, 0.25, 1.0 - 0.25 - 0.25 - 0.25, 0.25};
    double du0 = 0.0, du1 = 0.0, du2 = 0.0;                  // [:165-170]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        du0 = du0 + valC[a] * g.gx[a];
        du1 = du1 + valC[a] * g.gy[a];
        du2 = du2 + valC[a] * g.gz[a];
    }
    const double force = -6.0;                               // [:172]
#pragma unroll
    for (int i = 0; i < 4; ++i) {                            // [:174-188]
        const double b1 = g.gx[i] * dvol;
        const double b2 = g.gy[i] * dvol;
        const double b3 = g.gz[i] * dvol;
        const double b4 = N[i] * dvol;
        double f = 0.0 + b4 * force;
        f = f - b1 * du0 - b2 * du1 - b3 * du2;
        F[i] = f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            K[i + 4 * j] = 0.0 + af * (b1 * (kx * g.gx[j]) + b2 * (ky * g.gy[j]) + b3 * (kz * g.gz[j]));
    }
    return true;
}

// The part of poisson_tet() one NODE of the element needs (gather assembly): column a and, when
// `need_row`, row a of Klocal plus Flocal(a) -- the same expressions in the same order, entry by
// entry, so the bits equal those of the full routine (a is a run-time index: selects, no branches).
PFEM_HD bool poisson_tet_node(const double x[4], const double y[4], const double z[4], double kx,
                              double ky, double kz, double af, const double valC[4], int a, bool need_row,
                              double Kcol[4], double Krow[4], double &Fa)
{
    TetGeom g;
    tet_geometry(x, y, z, g);
    if (g.jac < 0.0) return false;
    const double dvol = kGaussWtTet * g.jac;
    const double N[4] = {0.25, 0.25, 1.0 - 0.25 - 0.25 - 0.25, 0.25};
    double du0 = 0.0, du1 = 0.0, du2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        du0 = du0 + valC[i] * g.gx[i];
        du1 = du1 + valC[i] * g.gy[i];
        du2 = du2 + valC[i] * g.gz[i];
    }
    double gxa = g.gx[0], gya = g.gy[0], gza = g.gz[0], Na = N[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i == a) { gxa = g.gx[i]; gya = g.gy[i]; gza = g.gz[i]; Na = N[i]; }
    const double force = -6.0;
    const double b1a = gxa * dvol, b2a = gya * dvol, b3a = gza * dvol, b4a = Na * dvol;
    double f = 0.0 + b4a * force;
    f = f - b1a * du0 - b2a * du1 - b3a * du2;
    Fa = f;
    const double cx = kx * gxa, cy = ky * gya, cz = kz * gza;
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // Klocal(j,a): b's of row j, gradient of column a
        const double b1 = g.gx[j] * dvol, b2 = g.gy[j] * dvol, b3 = g.gz[j] * dvol;
        Kcol[j] = 0.0 + af * (b1 * cx + b2 * cy + b3 * cz);
    }
    if (need_row) {
#pragma unroll
        for (int j = 0; j < 4; ++j)        // Klocal(a,j)
            Krow[j] = 0.0 + af * (b1a * (kx * g.gx[j]) + b2a * (ky * g.gy[j]) + b3a * (kz * g.gz[j]));
    }
    return true;
}

// poisson_tet_node() on the lean geometry for valC = 0 (what the drivers pass, and all the gather kernel ever sees): the
// du terms vanish (b * (+0) = +-0, f - (+-0) = f) and the `0.0 +` of the literal form only turns a -0 into +0 -- again the
// same bits for every nonzero result (see tet_geometry_lean).
PFEM_HD bool poisson_tet_node_lean(const double x[4], const double y[4], const double z[4], double kx, double ky, double kz,
                                   double af, int a, bool need_row, double Kcol[4], double Krow[4], double &Fa)
{
    TetGeom g;
#ifdef PFEM_LAB_FREE_GEOMETRY
    // lab build only (make LAB=1 EXTRA=-DPFEM_LAB_FREE_GEOMETRY, tools/r06/geom.sh): what the gather kernel would cost if an
    // element's Jacobian, its inverse and the gradients came for nothing -- the upper bound of any scheme that shares them
    // between the element's four visits.  WRONG matrix, timing only.
    g.jac = x[1] - x[0] + 1.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { g.gx[i] = y[i]; g.gy[i] = z[i]; g.gz[i] = x[i]; }
#else
    tet_geometry_lean(x, y, z, g);
#endif
    if (g.jac < 0.0) return false;
    const double dvol = kGaussWtTet * g.jac;
    double gxa = g.gx[0], gya = g.gy[0], gza = g.gz[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i == a) { gxa = g.gx[i]; gya = g.gy[i]; gza = g.gz[i]; }
    const double Na = a == 2 ? 1.0 - 0.25 - 0.25 - 0.25 : 0.25;
    const double b1a = gxa * dvol, b2a = gya * dvol, b3a = gza * dvol;
    Fa = (Na * dvol) * -6.0;
    const double cx = kx * gxa, cy = ky * gya, cz = kz * gza;
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // Klocal(j,a)
        const double b1 = g.gx[j] * dvol, b2 = g.gy[j] * dvol, b3 = g.gz[j] * dvol;
        Kcol[j] = af * (b1 * cx + b2 * cy + b3 * cz);
    }
    if (need_row) {
#pragma unroll
        for (int j = 0; j < 4; ++j)        // Klocal(a,j)
            Krow[j] = af * (b1a * (kx * g.gx[j]) + b2a * (ky * g.gy[j]) + b3a * (kz * g.gz[j]));
    }
    return true;
}

// ---------------------------------------------------------------------------
// P1 triangle [elementutilitiesbasisfuncs.F:16-52,165-234]: N=(xi3,xi1,xi2),
// parametric gradients (-1,-1),(1,0),(0,1); mapping rows n2-n1, n3-n1.
// ---------------------------------------------------------------------------
PFEM_HD bool poisson_tria(const double x[3], const double y[3], double kx, double ky, double af,
                          const double valC[3], double K[9], double F[3])
{
    const double u1[3] = {-1.0, 1.0, 0.0};
    const double u2[3] = {-1.0, 0.0, 1.0};
    double b11 = 0.0, b21 = 0.0, b12 = 0.0, b22 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {                            // [:207-215]
        b11 = b11 + (x[a] * u1[a]);
        b21 = b21 + (x[a] * u2[a]);
        b12 = b12 + (y[a] * u1[a]);
        b22 = b22 + (y[a] * u2[a]);
    }
    const double jac = b11 * b22 - b12 * b21;                // [:217]
    const double di = 1.0 / jac;
    const double i11 = b22 * di, i12 = -b12 * di, i21 = -b21 * di, i22 = b11 * di;  // [:221-224]
    double gx[3], gy[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gx[a] = u1[a] * i11 + u2[a] * i12;
        gy[a] = u1[a] * i21 + u2[a] * i22;
    }
    if (jac < 0.0) return false;                             // elementutilitiespoisson.F:72
    const double xi1 = kGaussPtTria, xi2 = kGaussPtTria;     // [:57]
    const double N[3] = {1.0 - xi1 - xi2, xi1, xi2};
    const double dvol = 0.5 * jac;                           // [:58,76]
    double du0 = 0.0, du1 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        du0 = du0 + valC[a] * gx[a];
        du1 = du1 + valC[a] * gy[a];
    }
    const double force = 0.0;                                // [:84]
#pragma unroll
    for (int i = 0; i < 3; ++i) {                            // [:86-97]
        const double b1 = gx[i] * dvol, b2 = gy[i] * dvol, b4 = N[i] * dvol;
        F[i] = 0.0 + b4 * force - b1 * du0 - b2 * du1;
#pragma unroll
        for (int j = 0; j < 3; ++j) K[i + 3 * j] = 0.0 + af * (b1 * (kx * gx[j]) + b2 * (ky * gy[j]));
    }
    return true;
}

// Inline element of the serial driver: Ke = area * (B B^T), B = [y_jk, x_kj]/(2 area),
// no load vector [triapoissonserialimpl1.F:573-594].
PFEM_HD bool poisson_tria_inline(const double x[3], const double y[3], double K[9], double F[3])
{
    const double area =
        0.5 * (x[1] * y[2] - x[2] * y[1] + x[2] * y[0] - x[0] * y[2] + x[0] * y[1] - x[1] * y[0]);
    const double two_a = 2.0 * area;
    const double bm[3][2] = {{(y[1] - y[2]) / two_a, (x[2] - x[1]) / two_a},
                             {(y[2] - y[0]) / two_a, (x[0] - x[2]) / two_a},
                             {(y[0] - y[1]) / two_a, (x[1] - x[0]) / two_a}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            K[i + 3 * j] = area * ((0.0 + bm[i][0] * bm[j][0]) + bm[i][1] * bm[j][1]);
    F[0] = F[1] = F[2] = 0.0;
    return true;
}

// ---------------------------------------------------------------------------
// Plane-stress elasticity on a P1 triangle [elementutilitieselasticity2D.F:23-153], the 2-D sibling
// (SURVEY 8f.1).  D = b1*[[1,nu,0],[nu,1,0],[0,0,(1-nu)]] with b1 = E/(1-nu^2) -- the reference's
// D(3,3) = b1*(1-nu), i.e. 2G with engineering shear, is reproduced as is.
// B (3x6) has columns (gx,0,gy) / (0,gy,gx); with the structural zeros dropped the reference's two
// MATMULs collapse, in the same summation order, to two products per entry:
//   K(2a+0,2b+0) = dvol*(ax*(D11 bx) + ay*(D33 by))   K(2a+0,2b+1) = dvol*(ax*(D12 by) + ay*(D33 bx))
//   K(2a+1,2b+0) = dvol*(ay*(D21 bx) + ax*(D33 by))   K(2a+1,2b+1) = dvol*(ay*(D22 by) + ax*(D33 bx))
// ---------------------------------------------------------------------------
struct TriaGeom {
    double gx[3], gy[3], jac;
};

PFEM_HD void tria_geometry(const double x[3], const double y[3], TriaGeom &g)
{
    const double u1[3] = {-1.0, 1.0, 0.0};
    const double u2[3] = {-1.0, 0.0, 1.0};
    double b11 = 0.0, b21 = 0.0, b12 = 0.0, b22 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {                            // elementutilitiesbasisfuncs.F:207-215
        b11 = b11 + (x[a] * u1[a]);
        b21 = b21 + (x[a] * u2[a]);
        b12 = b12 + (y[a] * u1[a]);
        b22 = b22 + (y[a] * u2[a]);
    }
    g.jac = b11 * b22 - b12 * b21;
    const double di = 1.0 / g.jac;
    const double i11 = b22 * di, i12 = -b12 * di, i21 = -b21 * di, i22 = b11 * di;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.gx[a] = u1[a] * i11 + u2[a] * i12;
        g.gy[a] = u1[a] * i21 + u2[a] * i22;
    }
}

struct Elast2dMat {
    double d11, d12, d33;
};

PFEM_HD Elast2dMat elast2d_material(double E, double nu)
{
    const double b1 = E / (1.0 - nu * nu);                   // [:60]
    Elast2dMat m;
    m.d11 = b1;
    m.d12 = b1 * nu;
    m.d33 = b1 * (1.0 - nu);
    return m;
}

// 2x2 block K(2a+p, 2b+q) from grad N_a = (ax,ay), grad N_b = (bx,by)
PFEM_HD void elast2d_block_v(double ax, double ay, double bx, double by, const Elast2dMat &m, double dvol,
                             double blk[2][2])
{
    blk[0][0] = dvol * (ax * (m.d11 * bx) + ay * (m.d33 * by));
    blk[0][1] = dvol * (ax * (m.d12 * by) + ay * (m.d33 * bx));
    blk[1][0] = dvol * (ay * (m.d12 * bx) + ax * (m.d33 * by));
    blk[1][1] = dvol * (ay * (m.d11 * by) + ax * (m.d33 * bx));
}

// shape functions at the Gauss point (1/3,1/3) given as REAL(4) literals [:73]: N = (xi3, xi1, xi2)
PFEM_HD void tria_shape_gp(double N[3])
{
    const double xi1 = kGaussPtTria, xi2 = kGaussPtTria;
    N[0] = 1.0 - xi1 - xi2;
    N[1] = xi1;
    N[2] = xi2;
}

// Full 6x6 (column-major) + load vector; elemData = (E, nu, thick, bx, by)
PFEM_HD bool elast_tria(const double x[3], const double y[3], double E, double nu, double thick,
                        const double bforce[2], double K[36], double F[6])
{
    TriaGeom g;
    tria_geometry(x, y, g);
    if (g.jac < 0.0) return false;                           // [:88]
    const double dvol = 0.5 * (g.jac * thick);               // [:92]
    const Elast2dMat m = elast2d_material(E, nu);
    double blk[2][2], N[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            elast2d_block_v(g.gx[a], g.gy[a], g.gx[b], g.gy[b], m, dvol, blk);
            K[(2 * a) + 6 * (2 * b)] = blk[0][0];
            K[(2 * a) + 6 * (2 * b + 1)] = blk[0][1];
            K[(2 * a + 1) + 6 * (2 * b)] = blk[1][0];
            K[(2 * a + 1) + 6 * (2 * b + 1)] = blk[1][1];
        }
    tria_shape_gp(N);
#pragma unroll
    for (int a = 0; a < 3; ++a) {                            // [:140-148]
        const double b4 = dvol * N[a];
        F[2 * a] = 0.0 + b4 * bforce[0];
        F[2 * a + 1] = 0.0 + b4 * bforce[1];
    }
    return true;
}

// ---------------------------------------------------------------------------
// Linear elasticity on a P1 tet [elementutilitieselasticity3D.F:248-393], intended
// semantics (one Gauss point, Ke = dvol * B^T (D B); DESIGN.md "deviations").
//
// B is 6x12 with Voigt rows [xx,yy,zz,xy,yz,zx] [:359-371] and D is the isotropic
// matrix of [:287-296].  The reference forms DB = MATMUL(D,B) and K = MATMUL(B^T,DB),
// both summed over the inner index ascending from zero.  Because B and D are sparse
// with structural zeros, and x + 0*y == x, 0 + x == x in IEEE arithmetic (finite y),
// the sums collapse to the few terms below IN THE SAME ORDER, i.e. the same bits --
// without ever materialising the 6x12 / 12x12 dense operands.
//   DB(:, 3a+0) = [D11 gx, D21 gx, D31 gx, G gy, 0, G gz]
//   DB(:, 3a+1) = [D12 gy, D22 gy, D32 gy, G gx, G gz, 0]
//   DB(:, 3a+2) = [D13 gz, D23 gz, D33 gz, 0, G gy, G gx]
// K(3a+p, 3b+q) = dvol * sum_k B(k,3a+p) DB(k,3b+q), k ascending over the nonzero rows
// of column 3a+p: p=0 -> k={0,3,5}, p=1 -> k={1,3,4}, p=2 -> k={2,4,5}.
// The callback form lets the device kernel stream 3x3 node blocks straight to the
// scatter without holding the 12x12 matrix in registers.
// ---------------------------------------------------------------------------
struct ElastMat {
    double d11, d12, gsh;  // b1(1-nu), b1 nu, b1 b2
};

PFEM_HD ElastMat elast_material(double E, double nu)
{
    const double b1 = E / ((1.0 + nu) * (1.0 - 2.0 * nu));  // [:284]
    const double b2 = (1.0 - 2.0 * nu) / 2.0;               // [:285]
    ElastMat m;
    m.d11 = b1 * (1.0 - nu);
    m.d12 = b1 * nu;
    m.gsh = b1 * b2;
    return m;
}

// 3x3 block K(3a+p,3b+q), p,q=0..2, into blk[p][q].
// Value form: (ax,ay,az) = grad N_a, (bx,by,bz) = grad N_b.
PFEM_HD void elast_block_v(double ax, double ay, double az, double bx, double by, double bz,
                           const ElastMat &m, double dvol, double blk[3][3]);

PFEM_HD void elast_block(const TetGeom &g, const ElastMat &m, double dvol, int a, int b,
                         double blk[3][3])
{
    elast_block_v(g.gx[a], g.gy[a], g.gz[a], g.gx[b], g.gy[b], g.gz[b], m, dvol, blk);
}

PFEM_HD void elast_block_v(double ax, double ay, double az, double bx, double by, double bz,
                           const ElastMat &m, double dvol, double blk[3][3])
{
    // DB columns of node b (rows 0..5); structural zeros omitted
    const double c0_0 = m.d11 * bx, c0_1 = m.d12 * bx, c0_2 = m.d12 * bx, c0_3 = m.gsh * by, c0_5 = m.gsh * bz;
    const double c1_0 = m.d12 * by, c1_1 = m.d11 * by, c1_2 = m.d12 * by, c1_3 = m.gsh * bx, c1_4 = m.gsh * bz;
    const double c2_0 = m.d12 * bz, c2_1 = m.d12 * bz, c2_2 = m.d11 * bz, c2_4 = m.gsh * by, c2_5 = m.gsh * bx;
    // row p=0 of node a: B rows {0:ax, 3:ay, 5:az}
    blk[0][0] = dvol * ((ax * c0_0 + ay * c0_3) + az * c0_5);
    blk[0][1] = dvol * (ax * c1_0 + ay * c1_3);
    blk[0][2] = dvol * (ax * c2_0 + az * c2_5);
    // row p=1: B rows {1:ay, 3:ax, 4:az}
    blk[1][0] = dvol * (ay * c0_1 + ax * c0_3);
    blk[1][1] = dvol * ((ay * c1_1 + ax * c1_3) + az * c1_4);
    blk[1][2] = dvol * (ay * c2_1 + az * c2_4);
    // row p=2: B rows {2:az, 4:ay, 5:ax}
    blk[2][0] = dvol * (az * c0_2 + ax * c0_5);
    blk[2][1] = dvol * (az * c1_2 + ay * c1_4);
    blk[2][2] = dvol * ((az * c2_2 + ay * c2_4) + ax * c2_5);
}

// Full 12x12 (column-major) + load vector; used by the host per-element entry point
// and by pfem_eval_elems.
PFEM_HD bool elast_tet(const double x[4], const double y[4], const double z[4], double E, double nu,
                       const double bforce[3], double K[144], double F[12])
{
    TetGeom g;
    tet_geometry(x, y, z, g);
    if (g.jac < 0.0) return false;                           // [:320]
    const double dvol = kGaussWtTet * g.jac;                 // [:305,324]
    const ElastMat m = elast_material(E, nu);
    double blk[3][3];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            elast_block(g, m, dvol, a, b, blk);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) K[(3 * a + p) + 12 * (3 * b + q)] = blk[p][q];
        }
    const double N[4] = {0.25, 0.25, 1.0 - 0.25 - 0.25 - 0.25, 0.25};
#pragma unroll
    for (int a = 0; a < 4; ++a) {                            // [:380-390]
        const double b4 = dvol * N[a];
        F[3 * a + 0] = 0.0 + b4 * bforce[0];
        F[3 * a + 1] = 0.0 + b4 * bforce[1];
        F[3 * a + 2] = 0.0 + b4 * bforce[2];
    }
    return true;
}

}  // namespace pfem
