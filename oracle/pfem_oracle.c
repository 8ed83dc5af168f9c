/*
 * pfem_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded-by-default CPU restatement of the PFEMFort implicit
 * FEM hot path (element stiffness -> Dirichlet lifting -> CSR assembly ->
 * Jacobi-preconditioned CG).  It exists to CHECK the HIP path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Every routine cites the reference file:line it restates (paths relative to
 * the PFEMFort checkout).  Floating-point operation ORDER follows the Fortran
 * source statement by statement and the file must be compiled with
 * -ffp-contract=off (the reference is built by gfortran -O3 for baseline
 * x86-64, which has no FMA), so element matrices agree BIT-FOR-BIT with the
 * flang-compiled reference routines in oracle/_ref (see tests/test_oracle_*).
 *
 * Pinning: element routines are pinned against the reference's own Fortran
 * compiled in place (oracle/_ref) and the golden vectors generated from it
 * (tests/golden); the mesh generator against the shipped input/tet10-* and
 * input/tet100-DirichBC files; the KSP boundary (PETSc is a third-party
 * dependency absent from the reference tree) is "parity unpinned" by the
 * reference and is pinned by analytic known answers + a direct sparse solve.
 *
 * Layout conventions (same as the Fortran drivers, column-major):
 *   coords(nNode,ndim)        -> SoA  x[nNode] | y[nNode] | z[nNode]
 *   elemNodeConn(nElem,npElem)-> SoA  conn[i*nElem + e]        (1-based in files,
 *                                      0-based everywhere in this file)
 *   ElemDofArray(nElem,nsize) -> SoA  edof[i*nElem + e], -1 = Dirichlet
 *   Klocal(nsize,nsize)       -> column-major K[i + nsize*j]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_ARG 1
#define ORC_ERR_NEG_JAC 3
#define ORC_ERR_NOMEM 6
#define ORC_ERR_PATTERN 8

/* element kinds (shared with the product header include/pfem_amd.h) */
#define ORC_POISSON_TRIA 1
#define ORC_POISSON_TET 2
#define ORC_ELAST_TET 3
#define ORC_POISSON_TRIA_INLINE 4
#define ORC_ELAST_TRIA 5

/* ------------------------------------------------------------------------ */
/* Element routines                                                          */
/* ------------------------------------------------------------------------ */

/* elementutilitiesbasisfuncs.F:242-289 (LagrangeBasisFunsTet, degree 1) and
 * :430-538 (computeBasisFunctions3D, ETYPE=4). */
static void basis_tet(const double xi[3], const double *xN, const double *yN,
                      const double *zN, double N[4], double dNdx[4],
                      double dNdy[4], double dNdz[4], double *Jac)
{
    double du1[4], du2[4], du3[4];
    double B[3][3], Binv[3][3], detinv;
    int ii;

    /* :265-281 */
    N[0] = xi[0];
    N[1] = xi[1];
    N[2] = 1.0 - xi[0] - xi[1] - xi[2];
    N[3] = xi[2];
    du1[0] = 1.0;  du1[1] = 0.0;  du1[3] = 0.0;  du1[2] = -1.0;
    du2[0] = 0.0;  du2[1] = 1.0;  du2[3] = 0.0;  du2[2] = -1.0;
    du3[0] = 0.0;  du3[1] = 0.0;  du3[3] = 1.0;  du3[2] = -1.0;

    /* :491-509  B(i,j) = sum_a coord_j(a) * dN_a/dxi_i */
    memset(B, 0, sizeof B);
    for (ii = 0; ii < 4; ++ii) {
        const double xx = xN[ii], yy = yN[ii], zz = zN[ii];
        B[0][0] = B[0][0] + (xx * du1[ii]);
        B[1][0] = B[1][0] + (xx * du2[ii]);
        B[2][0] = B[2][0] + (xx * du3[ii]);
        B[0][1] = B[0][1] + (yy * du1[ii]);
        B[1][1] = B[1][1] + (yy * du2[ii]);
        B[2][1] = B[2][1] + (yy * du3[ii]);
        B[0][2] = B[0][2] + (zz * du1[ii]);
        B[1][2] = B[1][2] + (zz * du2[ii]);
        B[2][2] = B[2][2] + (zz * du3[ii]);
    }
    /* :512-514 */
    *Jac = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]);
    *Jac = *Jac + B[0][1] * (B[1][2] * B[2][0] - B[1][0] * B[2][2]);
    *Jac = *Jac + B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
    /* :517 */
    detinv = 1.0 / *Jac;
    /* :520-528 */
    Binv[0][0] = +detinv * (B[1][1] * B[2][2] - B[1][2] * B[2][1]);
    Binv[1][0] = -detinv * (B[1][0] * B[2][2] - B[1][2] * B[2][0]);
    Binv[2][0] = +detinv * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
    Binv[0][1] = -detinv * (B[0][1] * B[2][2] - B[0][2] * B[2][1]);
    Binv[1][1] = +detinv * (B[0][0] * B[2][2] - B[0][2] * B[2][0]);
    Binv[2][1] = -detinv * (B[0][0] * B[2][1] - B[0][1] * B[2][0]);
    Binv[0][2] = +detinv * (B[0][1] * B[1][2] - B[0][2] * B[1][1]);
    Binv[1][2] = -detinv * (B[0][0] * B[1][2] - B[0][2] * B[1][0]);
    Binv[2][2] = +detinv * (B[0][0] * B[1][1] - B[0][1] * B[1][0]);
    /* :532-536 */
    for (ii = 0; ii < 4; ++ii) {
        dNdx[ii] = du1[ii] * Binv[0][0] + du2[ii] * Binv[0][1] + du3[ii] * Binv[0][2];
        dNdy[ii] = du1[ii] * Binv[1][0] + du2[ii] * Binv[1][1] + du3[ii] * Binv[1][2];
        dNdz[ii] = du1[ii] * Binv[2][0] + du2[ii] * Binv[2][1] + du3[ii] * Binv[2][2];
    }
}

/* elementutilitiesbasisfuncs.F:16-52 (LagrangeBasisFunsTria, degree 1) and
 * :165-234 (computeBasisFunctions2D, ETYPE=1). */
static void basis_tria(const double xi[2], const double *xN, const double *yN,
                       double N[3], double dNdx[3], double dNdy[3], double *Jac)
{
    double du1[3], du2[3], B[2][2], Binv[2][2], detinv, xi3;
    int ii;
    xi3 = 1.0 - xi[0] - xi[1];                       /* :30 */
    N[0] = xi3;  N[1] = xi[0];  N[2] = xi[1];        /* :43-45 */
    du1[0] = -1.0;  du1[1] = 1.0;  du1[2] = 0.0;     /* :47-49 */
    du2[0] = -1.0;  du2[1] = 0.0;  du2[2] = 1.0;     /* :51-53 */
    memset(B, 0, sizeof B);
    for (ii = 0; ii < 3; ++ii) {                     /* :207-215 */
        const double xx = xN[ii], yy = yN[ii];
        B[0][0] = B[0][0] + (xx * du1[ii]);
        B[1][0] = B[1][0] + (xx * du2[ii]);
        B[0][1] = B[0][1] + (yy * du1[ii]);
        B[1][1] = B[1][1] + (yy * du2[ii]);
    }
    *Jac = B[0][0] * B[1][1] - B[0][1] * B[1][0];    /* :217 */
    detinv = 1.0 / *Jac;                             /* :219 */
    Binv[0][0] = B[1][1] * detinv;                   /* :221-224 */
    Binv[0][1] = -B[0][1] * detinv;
    Binv[1][0] = -B[1][0] * detinv;
    Binv[1][1] = B[0][0] * detinv;
    for (ii = 0; ii < 3; ++ii) {                     /* :227-230 */
        dNdx[ii] = du1[ii] * Binv[0][0] + du2[ii] * Binv[0][1];
        dNdy[ii] = du1[ii] * Binv[1][0] + du2[ii] * Binv[1][1];
    }
}

/* elementutilitiespoisson.F:107-193 StiffnessResidualPoissonLinearTetra.
 * elemData = (kx,ky,kz), timeData(2)=af, valC(4); K col-major 4x4. */
int orc_poisson_tet_ke(const double *xN, const double *yN, const double *zN,
                       const double *elemData, const double *timeData,
                       const double *valC, double *K, double *F)
{
    const double kx = elemData[0], ky = elemData[1], kz = elemData[2];
    const double af = timeData[1];
    /* REAL(4) literal 1.0/6.0 widened to double (:142; SURVEY finding 4) */
    const double gwts = (double)(1.0f / 6.0f);
    const double xi[3] = {0.25, 0.25, 0.25};          /* :141 */
    double N[4], dNdx[4], dNdy[4], dNdz[4], Jac, dvol, du[3], force;
    double b1, b2, b3, b4;
    int ii, jj;

    for (ii = 0; ii < 16; ++ii) K[ii] = 0.0;          /* :146 */
    for (ii = 0; ii < 4; ++ii) F[ii] = 0.0;
    basis_tet(xi, xN, yN, zN, N, dNdx, dNdy, dNdz, &Jac);   /* :153 */
    if (Jac < 0.0) return ORC_ERR_NEG_JAC;            /* :157 STOP */
    dvol = gwts * Jac;                                /* :161 */
    du[0] = du[1] = du[2] = 0.0;                      /* :165-170 */
    for (ii = 0; ii < 4; ++ii) {
        du[0] = du[0] + valC[ii] * dNdx[ii];
        du[1] = du[1] + valC[ii] * dNdy[ii];
        du[2] = du[2] + valC[ii] * dNdz[ii];
    }
    force = -6.0;                                     /* :172 */
    for (ii = 0; ii < 4; ++ii) {                      /* :174-188 */
        b1 = dNdx[ii] * dvol;
        b2 = dNdy[ii] * dvol;
        b3 = dNdz[ii] * dvol;
        b4 = N[ii] * dvol;
        F[ii] = F[ii] + b4 * force;
        F[ii] = F[ii] - b1 * du[0] - b2 * du[1] - b3 * du[2];
        for (jj = 0; jj < 4; ++jj) {
            K[ii + 4 * jj] = K[ii + 4 * jj] +
                af * (b1 * (kx * dNdx[jj]) + b2 * (ky * dNdy[jj]) + b3 * (kz * dNdz[jj]));
        }
    }
    return ORC_OK;
}

/* elementutilitiespoisson.F:23-101 StiffnessResidualPoissonLinearTria. */
int orc_poisson_tria_ke(const double *xN, const double *yN,
                        const double *elemData, const double *timeData,
                        const double *valC, double *K, double *F)
{
    const double kx = elemData[0], ky = elemData[1];
    const double af = timeData[1];
    /* REAL(4) literal 1.0/3.0 (:57) */
    const double xi[2] = {(double)(1.0f / 3.0f), (double)(1.0f / 3.0f)};
    const double gwts = 0.5;                          /* :58 */
    double N[3], dNdx[3], dNdy[3], Jac, dvol, du[2], force, b1, b2, b4;
    int ii, jj;

    for (ii = 0; ii < 9; ++ii) K[ii] = 0.0;
    for (ii = 0; ii < 3; ++ii) F[ii] = 0.0;
    basis_tria(xi, xN, yN, N, dNdx, dNdy, &Jac);      /* :68 */
    if (Jac < 0.0) return ORC_ERR_NEG_JAC;            /* :72 */
    dvol = gwts * Jac;                                /* :76 */
    du[0] = du[1] = 0.0;
    for (ii = 0; ii < 3; ++ii) {                      /* :78-82 */
        du[0] = du[0] + valC[ii] * dNdx[ii];
        du[1] = du[1] + valC[ii] * dNdy[ii];
    }
    force = 0.0;                                      /* :84 */
    for (ii = 0; ii < 3; ++ii) {                      /* :86-97 */
        b1 = dNdx[ii] * dvol;
        b2 = dNdy[ii] * dvol;
        b4 = N[ii] * dvol;
        F[ii] = F[ii] + b4 * force - b1 * du[0] - b2 * du[1];
        for (jj = 0; jj < 3; ++jj)
            K[ii + 3 * jj] = K[ii + 3 * jj] + af * (b1 * (kx * dNdx[jj]) + b2 * (ky * dNdy[jj]));
    }
    return ORC_OK;
}

/* triapoissonserialimpl1.F:573-594: inline Ke = area * B * B^T (no source). */
int orc_poisson_tria_inline_ke(const double *xN, const double *yN, double *K, double *F)
{
    const double x1 = xN[0], x2 = xN[1], x3 = xN[2];
    const double y1 = yN[0], y2 = yN[1], y3 = yN[2];
    double area, Bm[3][2], s;
    int i, j, k;
    area = 0.5 * (x2 * y3 - x3 * y2 + x3 * y1 - x1 * y3 + x1 * y2 - x2 * y1);   /* :580 */
    Bm[0][0] = y2 - y3;  Bm[1][0] = y3 - y1;  Bm[2][0] = y1 - y2;    /* :583-587 */
    Bm[0][1] = x3 - x2;  Bm[1][1] = x1 - x3;  Bm[2][1] = x2 - x1;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 2; ++j) Bm[i][j] = Bm[i][j] / (2.0 * area);  /* :589 */
    for (i = 0; i < 3; ++i)                                          /* :593-594 */
        for (j = 0; j < 3; ++j) {
            s = 0.0;
            for (k = 0; k < 2; ++k) s = s + Bm[i][k] * Bm[j][k];
            K[i + 3 * j] = area * s;
        }
    F[0] = F[1] = F[2] = 0.0;                                        /* :567 */
    return ORC_OK;
}

/* elementutilitieselasticity3D.F:248-393 StiffnessResidualElasticityLinearTetra
 * with the INTENDED semantics documented in SURVEY finding 5 / A.3#1: one
 * Gauss point (1/4,1/4,1/4), ETYPE=4 basis, Ke = dvol * B^T (D B).
 * elemData = (E, nu, thick, bx, by, bz). K col-major 12x12. */
int orc_elast_tet_ke(const double *xN, const double *yN, const double *zN,
                     const double *elemData, const double *timeData,
                     double *K, double *F)
{
    const double E = elemData[0], nu = elemData[1];
    const double bforce[3] = {elemData[3], elemData[4], elemData[5]};
    const double gwts = (double)(1.0f / 6.0f);        /* :305 */
    const double xi[3] = {0.25, 0.25, 0.25};
    double D[6][6], Bm[6][12], DB[6][12], N[4], dNdx[4], dNdy[4], dNdz[4];
    double Jac, dvol, b1, b2, b4, s;
    int i, j, k, ii, TI;
    (void)timeData;

    b1 = E / ((1.0 + nu) * (1.0 - 2.0 * nu));         /* :284 */
    b2 = (1.0 - 2.0 * nu) / 2.0;                      /* :285 */
    memset(D, 0, sizeof D);                           /* :287-296 */
    D[0][0] = b1 * (1.0 - nu);  D[0][1] = b1 * nu;          D[0][2] = b1 * nu;
    D[1][0] = b1 * nu;          D[1][1] = b1 * (1.0 - nu);  D[1][2] = b1 * nu;
    D[2][0] = b1 * nu;          D[2][1] = b1 * nu;          D[2][2] = b1 * (1.0 - nu);
    D[3][3] = b1 * b2;
    D[4][4] = b1 * b2;
    D[5][5] = b1 * b2;

    for (i = 0; i < 144; ++i) K[i] = 0.0;             /* :309 */
    for (i = 0; i < 12; ++i) F[i] = 0.0;
    basis_tet(xi, xN, yN, zN, N, dNdx, dNdy, dNdz, &Jac);
    if (Jac < 0.0) return ORC_ERR_NEG_JAC;            /* :320 */
    dvol = gwts * Jac;                                /* :324 */

    memset(Bm, 0, sizeof Bm);                         /* :357-371 */
    for (ii = 0; ii < 4; ++ii) {
        TI = ii * 3;
        Bm[0][TI] = dNdx[ii];
        Bm[1][TI + 1] = dNdy[ii];
        Bm[2][TI + 2] = dNdz[ii];
        Bm[3][TI] = dNdy[ii];  Bm[3][TI + 1] = dNdx[ii];
        Bm[4][TI + 1] = dNdz[ii];  Bm[4][TI + 2] = dNdy[ii];
        Bm[5][TI] = dNdz[ii];  Bm[5][TI + 2] = dNdx[ii];
    }
    /* :374  Bmat = MATMUL(Dmat, Bmat): k ascending from 0.0 */
    for (i = 0; i < 6; ++i)
        for (j = 0; j < 12; ++j) {
            s = 0.0;
            for (k = 0; k < 6; ++k) s = s + D[i][k] * Bm[k][j];
            DB[i][j] = s;
        }
    /* :376-377  Klocal = dvol * MATMUL(TRANSPOSE(B), D B) */
    for (i = 0; i < 12; ++i)
        for (j = 0; j < 12; ++j) {
            s = 0.0;
            for (k = 0; k < 6; ++k) s = s + Bm[k][i] * DB[k][j];
            K[i + 12 * j] = dvol * s;
        }
    for (ii = 0; ii < 4; ++ii) {                      /* :380-390 */
        TI = ii * 3;
        b4 = dvol * N[ii];
        F[TI] = F[TI] + b4 * bforce[0];
        F[TI + 1] = F[TI + 1] + b4 * bforce[1];
        F[TI + 2] = F[TI + 2] + b4 * bforce[2];
    }
    return ORC_OK;
}

/* elementutilitieselasticity2D.F:23-153 StiffnessResidualElasticityLinearTria: plane stress
 * with the reference's D(3,3) = b1*(1-nu), one Gauss point (1/3,1/3) as REAL(4)
 * literals, weight 0.5, dvol = gwts*(Jac*thick).  elemData = (E, nu, thick, bx, by). K col-major 6x6. */
int orc_elast_tria_ke(const double *xN, const double *yN, const double *elemData,
                      const double *timeData, double *K, double *F)
{
    const double E = elemData[0], nu = elemData[1], thick = elemData[2];
    const double bforce[2] = {elemData[3], elemData[4]};
    const double xi[2] = {(double)(1.0f / 3.0f), (double)(1.0f / 3.0f)};   /* :73 */
    const double gwts = 0.5;
    double D[3][3], Bm[3][6], DB[3][6], N[3], dNdx[3], dNdy[3], Jac, dvol, b1, b4, s;
    int i, j, k, ii, TI;
    (void)timeData;
    b1 = E / (1.0 - nu * nu);                           /* :60 */
    D[0][0] = b1;       D[0][1] = b1 * nu;  D[0][2] = 0.0;   /* :63-65 */
    D[1][0] = b1 * nu;  D[1][1] = b1;       D[1][2] = 0.0;
    D[2][0] = 0.0;      D[2][1] = 0.0;      D[2][2] = b1 * (1.0 - nu);
    for (i = 0; i < 36; ++i) K[i] = 0.0;
    for (i = 0; i < 6; ++i) F[i] = 0.0;
    basis_tria(xi, xN, yN, N, dNdx, dNdy, &Jac);        /* :84 */
    if (Jac < 0.0) return ORC_ERR_NEG_JAC;              /* :88 */
    dvol = gwts * (Jac * thick);                        /* :92 */
    memset(Bm, 0, sizeof Bm);                           /* :124-131 */
    for (ii = 0; ii < 3; ++ii) {
        TI = ii * 2;
        Bm[0][TI] = dNdx[ii];  Bm[0][TI + 1] = 0.0;
        Bm[1][TI] = 0.0;       Bm[1][TI + 1] = dNdy[ii];
        Bm[2][TI] = dNdy[ii];  Bm[2][TI + 1] = dNdx[ii];
    }
    for (i = 0; i < 3; ++i)                             /* :134 Bmat = MATMUL(Dmat, Bmat) */
        for (j = 0; j < 6; ++j) {
            s = 0.0;
            for (k = 0; k < 3; ++k) s = s + D[i][k] * Bm[k][j];
            DB[i][j] = s;
        }
    for (i = 0; i < 6; ++i)                             /* :136-137 */
        for (j = 0; j < 6; ++j) {
            s = 0.0;
            for (k = 0; k < 3; ++k) s = s + Bm[k][i] * DB[k][j];
            K[i + 6 * j] = dvol * s;
        }
    for (ii = 0; ii < 3; ++ii) {                        /* :140-148 */
        TI = ii * 2;
        b4 = dvol * N[ii];
        F[TI] = F[TI] + b4 * bforce[0];
        F[TI + 1] = F[TI + 1] + b4 * bforce[1];
    }
    return ORC_OK;
}

/* Geometry of each element kind. */
static int kind_npelem(int kind) { return (kind == ORC_POISSON_TET || kind == ORC_ELAST_TET) ? 4 : 3; }
static int kind_ndof(int kind) { return kind == ORC_ELAST_TET ? 3 : (kind == ORC_ELAST_TRIA ? 2 : 1); }
static int kind_ndim(int kind) { return (kind == ORC_POISSON_TET || kind == ORC_ELAST_TET) ? 3 : 2; }

/* Evaluate one element of a mesh: gathers coordinates like the driver does
 * (tetrapoissonparallelimpl1.F:832-838; coords already in NEW numbering here)
 * and calls the element routine. valC = 0 as in the drivers (:824). */
static int eval_elem(int kind, int64_t e, int64_t nElem, const int32_t *conn,
                     int64_t nNode, const double *xyz, const double *elemData,
                     const double *timeData, double *K, double *F)
{
    double xN[4], yN[4], zN[4], valC[12] = {0};
    const int np = kind_npelem(kind), nd = kind_ndim(kind);
    int i;
    for (i = 0; i < np; ++i) {
        const int64_t n = conn[(int64_t)i * nElem + e];
        xN[i] = xyz[n];
        yN[i] = xyz[nNode + n];
        zN[i] = nd == 3 ? xyz[2 * nNode + n] : 0.0;
    }
    switch (kind) {
    case ORC_POISSON_TRIA: return orc_poisson_tria_ke(xN, yN, elemData, timeData, valC, K, F);
    case ORC_POISSON_TRIA_INLINE: return orc_poisson_tria_inline_ke(xN, yN, K, F);
    case ORC_POISSON_TET: return orc_poisson_tet_ke(xN, yN, zN, elemData, timeData, valC, K, F);
    case ORC_ELAST_TET: return orc_elast_tet_ke(xN, yN, zN, elemData, timeData, K, F);
    case ORC_ELAST_TRIA: return orc_elast_tria_ke(xN, yN, elemData, timeData, K, F);
    }
    return ORC_ERR_ARG;
}

/* Batch evaluation: Kout[e*nsize*nsize ...] col-major per element, Fout[e*nsize]. */
int orc_eval_elems(int kind, int64_t nElem, const int32_t *conn, int64_t nNode,
                   const double *xyz, const double *elemData, const double *timeData,
                   double *Kout, double *Fout)
{
    const int nsize = kind_npelem(kind) * kind_ndof(kind);
    int64_t e;
    for (e = 0; e < nElem; ++e) {
        int rc = eval_elem(kind, e, nElem, conn, nNode, xyz, elemData, timeData,
                           Kout + e * nsize * nsize, Fout + e * nsize);
        if (rc) return rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* Integer bookkeeping (tetrapoissonparallelimpl1.F:316-367, 393-734)        */
/* ------------------------------------------------------------------------ */

/* Everything 0-based here; "+1" noted where the Fortran is 1-based.
 * Inputs : nNode, ndof, DirichBC triples (node 0-based, dof 0-based, value),
 *          nParts and node_proc_id (ignored when nParts==1).
 * Outputs: node_map_get_old[new]=old, node_map_get_new[old]=new,
 *          NodeDofArrayNew[new*ndof+d] = free dof id (0-based) or -1,
 *          solnApplied[new*ndof+d] (values re-entered at NEW ids, :668-677),
 *          node_start[p]/node_end[p] (exclusive end), row_start[p]/row_end[p]
 *          (exclusive end), returns size_global through *size_global. */
int orc_dof_numbering(int64_t nNode, int ndof, int64_t nDBC, const int32_t *dbc_node,
                      const int32_t *dbc_dof, const double *dbc_val, int nParts,
                      const int32_t *node_proc_id, int32_t *node_map_get_old,
                      int32_t *node_map_get_new, int32_t *NodeDofArrayNew,
                      double *solnApplied, int64_t *node_start, int64_t *node_end,
                      int64_t *row_start, int64_t *row_end, int64_t *size_global)
{
    int64_t ii, kk, ind;
    int jj, p;
    int8_t *NodeTypeOld = (int8_t *)calloc((size_t)(nNode * ndof), 1);
    if (!NodeTypeOld) return ORC_ERR_NOMEM;

    for (ii = 0; ii < nNode * ndof; ++ii) solnApplied[ii] = 0.0;   /* :328-338 */
    /* :341-355; solnApplied is indexed by OLD ids first ... */
    for (ii = 0; ii < nDBC; ++ii) {
        NodeTypeOld[(int64_t)dbc_node[ii] * ndof + dbc_dof[ii]] = 1;
        solnApplied[(int64_t)dbc_node[ii] * ndof + dbc_dof[ii]] = dbc_val[ii];
    }
    /* node maps */
    if (nParts == 1) {                                /* :402-421 */
        for (ii = 0; ii < nNode; ++ii) { node_map_get_old[ii] = (int32_t)ii; node_map_get_new[ii] = (int32_t)ii; }
        node_start[0] = 0;  node_end[0] = nNode;
    } else {                                          /* :500-595 */
        kk = 0;
        for (p = 0; p < nParts; ++p) {
            node_start[p] = kk;
            for (ii = 0; ii < nNode; ++ii)
                if (node_proc_id[ii] == p) node_map_get_old[kk++] = (int32_t)ii;
            node_end[p] = kk;
        }
        if (kk != nNode) { free(NodeTypeOld); return ORC_ERR_ARG; }
        for (ii = 0; ii < nNode; ++ii) node_map_get_new[node_map_get_old[ii]] = (int32_t)ii;
        /* ... then re-entered at NEW ids WITHOUT clearing the old slots (:668-677) */
        for (ii = 0; ii < nDBC; ++ii) {
            const int64_t n1 = node_map_get_new[dbc_node[ii]];
            solnApplied[n1 * ndof + dbc_dof[ii]] = dbc_val[ii];
        }
    }
    /* NodeDofArrayNew: free dofs numbered scanning NEW node order (:601-612) */
    ind = 0;
    for (ii = 0; ii < nNode; ++ii)
        for (jj = 0; jj < ndof; ++jj) {
            const int64_t old = node_map_get_old[ii];
            if (NodeTypeOld[old * ndof + jj] == 0) NodeDofArrayNew[ii * ndof + jj] = (int32_t)ind++;
            else NodeDofArrayNew[ii * ndof + jj] = -1;
        }
    *size_global = ind;
    /* row ranges per part (:622-636) */
    for (p = 0; p < nParts; ++p) {
        int64_t rs = -1, re = -1;
        for (ii = node_start[p]; ii < node_end[p]; ++ii)
            for (jj = 0; jj < ndof; ++jj) {
                const int32_t d = NodeDofArrayNew[ii * ndof + jj];
                if (d >= 0) { if (rs < 0) rs = d; re = d + 1; }
            }
        if (rs < 0) { rs = p ? row_end[p - 1] : 0; re = rs; }
        row_start[p] = rs;  row_end[p] = re;
    }
    free(NodeTypeOld);
    return ORC_OK;
}

/* ElemDofArray (:698-713) from conn in NEW numbering (0-based), SoA. */
int orc_elem_dof_array(int64_t nElem, int npElem, int ndof, const int32_t *conn_new,
                       const int32_t *NodeDofArrayNew, int32_t *edof)
{
    int64_t e;
    int i, j;
    for (e = 0; e < nElem; ++e)
        for (i = 0; i < npElem; ++i) {
            const int64_t n2 = conn_new[(int64_t)i * nElem + e];
            for (j = 0; j < ndof; ++j)
                edof[(int64_t)(i * ndof + j) * nElem + e] = NodeDofArrayNew[n2 * ndof + j];
        }
    return ORC_OK;
}

/* assyForSoln (:722-734): k-th free dof -> node*ndof+d (0-based). */
int orc_assy_for_soln(int64_t nNode, int ndof, const int32_t *NodeDofArrayNew, int32_t *assy)
{
    int64_t ii, count = 0;
    int jj;
    for (ii = 0; ii < nNode; ++ii)
        for (jj = 0; jj < ndof; ++jj)
            if (NodeDofArrayNew[ii * ndof + jj] >= 0) assy[count++] = (int32_t)(ii * ndof + jj);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* Sparsity pattern + assembly (PETSc MatSetValues semantics, third-party)   */
/* ------------------------------------------------------------------------ */

static int cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* Pattern of the INSERT_VALUES pass (tetrapoissonparallelimpl1.F:786-802):
 * every (row,col) pair of every element with both indices >= 0, columns sorted
 * ascending inside each row (PETSc AIJ after final assembly).
 * Two-call protocol: cols==NULL -> only rowptr (N+1) is filled. */
int orc_csr_pattern(int64_t nElem, int nsize, const int32_t *edof, int64_t N,
                    int64_t *rowptr, int32_t *cols)
{
    /* per-row candidate lists via counting sort of (row,col) keys by row */
    int64_t e, k, *cnt, *start, *keys, total = 0, r;
    int i, j;
    cnt = (int64_t *)calloc((size_t)N + 1, sizeof *cnt);
    if (!cnt) return ORC_ERR_NOMEM;
    for (e = 0; e < nElem; ++e)
        for (i = 0; i < nsize; ++i) {
            const int32_t row = edof[(int64_t)i * nElem + e];
            int nc = 0;
            if (row < 0) continue;
            for (j = 0; j < nsize; ++j) nc += edof[(int64_t)j * nElem + e] >= 0;
            cnt[row] += nc;
        }
    start = (int64_t *)malloc(((size_t)N + 1) * sizeof *start);
    if (!start) { free(cnt); return ORC_ERR_NOMEM; }
    for (r = 0; r < N; ++r) { start[r] = total; total += cnt[r]; }
    start[N] = total;
    keys = (int64_t *)malloc((size_t)(total ? total : 1) * sizeof *keys);
    if (!keys) { free(cnt); free(start); return ORC_ERR_NOMEM; }
    memset(cnt, 0, ((size_t)N + 1) * sizeof *cnt);
    for (e = 0; e < nElem; ++e)
        for (i = 0; i < nsize; ++i) {
            const int32_t row = edof[(int64_t)i * nElem + e];
            if (row < 0) continue;
            for (j = 0; j < nsize; ++j) {
                const int32_t col = edof[(int64_t)j * nElem + e];
                if (col >= 0) keys[start[row] + cnt[row]++] = col;
            }
        }
    /* sort + unique every row's candidates (rows are independent: threaded), then the row pointers, then the copy */
#pragma omp parallel for schedule(dynamic, 4096) private(k) if (N > 100000)
    for (r = 0; r < N; ++r) {
        int64_t *seg = keys + start[r], n = cnt[r], u = 0;
        qsort(seg, (size_t)n, sizeof *seg, cmp_i64);
        for (k = 0; k < n; ++k)
            if (k == 0 || seg[k] != seg[k - 1]) seg[u++] = seg[k];
        cnt[r] = u;
    }
    rowptr[0] = 0;
    for (r = 0; r < N; ++r) rowptr[r + 1] = rowptr[r] + cnt[r];
    if (cols) {
#pragma omp parallel for schedule(static) private(k) if (N > 100000)
        for (r = 0; r < N; ++r)
            for (k = 0; k < cnt[r]; ++k) cols[rowptr[r] + k] = (int32_t)keys[start[r] + k];
    }
    free(keys);  free(start);  free(cnt);
    return ORC_OK;
}

static inline int64_t csr_find(const int64_t *rowptr, const int32_t *cols, int64_t row, int32_t col)
{
    int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cols[mid] == col) return mid;
        if (cols[mid] < col) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* Serial assembly in ascending element order = the reference on one rank
 * (tetrapoissonparallelimpl1.F:828-884 / tetraelasticityparallelimpl1.F:906-965
 *  / triapoissonserialimpl1.F:559-650).  Only elements with
 * elem_proc_id[e]==part are processed when elem_proc_id != NULL.
 *   MatSetValues(ADD_VALUES): negative row/col ignored; the value block is
 *   read ROW-MAJOR by PETSc while Fortran passes Klocal column-major, i.e.
 *   entry (row_i, col_j) receives Klocal(j,i) -- restated literally here.
 *   Dirichlet lifting: Flocal(jj) -= Klocal(jj,ii)*solnApplied(...) (:859-870)
 *   VecSetValues(ADD_VALUES) with VEC_IGNORE_NEGATIVE_INDICES (solverpetsc.F:142).
 * solnApplied is indexed by NEW node*ndof+d (0-based). */
int orc_assemble(int kind, int64_t nElem, const int32_t *conn, int64_t nNode,
                 const double *xyz, const int32_t *edof, const double *solnApplied,
                 const double *elemData, const double *timeData,
                 const int32_t *elem_proc_id, int part,
                 int64_t N, const int64_t *rowptr, const int32_t *cols,
                 double *vals, double *rhs)
{
    const int np = kind_npelem(kind), ndof = kind_ndof(kind), nsize = np * ndof;
    double K[144], F[12];
    int32_t idx[12];
    int64_t e;
    int ii, jj, rc;
    (void)N;
    for (e = 0; e < nElem; ++e) {
        if (elem_proc_id && elem_proc_id[e] != part) continue;
        rc = eval_elem(kind, e, nElem, conn, nNode, xyz, elemData, timeData, K, F);
        if (rc) return rc;
        for (ii = 0; ii < nsize; ++ii) idx[ii] = edof[(int64_t)ii * nElem + e];
        /* MatSetValues, row-major read of the column-major block */
        for (ii = 0; ii < nsize; ++ii) {
            if (idx[ii] < 0) continue;
            for (jj = 0; jj < nsize; ++jj) {
                int64_t p;
                if (idx[jj] < 0) continue;
                p = csr_find(rowptr, cols, idx[ii], idx[jj]);
                if (p < 0) return ORC_ERR_PATTERN;
                vals[p] += K[jj + nsize * ii];
            }
        }
        /* Dirichlet lifting */
        for (ii = 0; ii < nsize; ++ii) {
            if (idx[ii] == -1) {
                const int64_t n = conn[(int64_t)(ii / ndof) * nElem + e];
                const double fact = solnApplied[n * ndof + ii % ndof];
                for (jj = 0; jj < nsize; ++jj)
                    if (idx[jj] != -1) F[jj] = F[jj] - K[jj + nsize * ii] * fact;
            }
        }
        for (ii = 0; ii < nsize; ++ii)
            if (idx[ii] >= 0) rhs[idx[ii]] += F[ii];
    }
    return ORC_OK;
}

/* Threaded variant for the CPU-baseline leg of bench.py ONLY (never for parity): the same element
 * loop split over OpenMP threads, every ADD_VALUES an atomic update.  This is what the reference's
 * "mpirun -np P" buys on a shared-memory node, minus the stash exchange; the sum order (hence the
 * last bits) depends on the thread interleaving, so tests always use orc_assemble. */
int orc_assemble_mt(int kind, int64_t nElem, const int32_t *conn, int64_t nNode,
                    const double *xyz, const int32_t *edof, const double *solnApplied,
                    const double *elemData, const double *timeData,
                    int64_t N, const int64_t *rowptr, const int32_t *cols,
                    double *vals, double *rhs)
{
    const int np = kind_npelem(kind), ndof = kind_ndof(kind), nsize = np * ndof;
    int err = ORC_OK;
    int64_t e;
    (void)N;
#pragma omp parallel for schedule(static)
    for (e = 0; e < nElem; ++e) {
        double K[144], F[12];
        int32_t idx[12];
        int ii, jj;
        int rc = eval_elem(kind, e, nElem, conn, nNode, xyz, elemData, timeData, K, F);
        if (rc) { err = rc; continue; }
        for (ii = 0; ii < nsize; ++ii) idx[ii] = edof[(int64_t)ii * nElem + e];
        for (ii = 0; ii < nsize; ++ii) {
            if (idx[ii] < 0) continue;
            for (jj = 0; jj < nsize; ++jj) {
                int64_t p;
                if (idx[jj] < 0) continue;
                p = csr_find(rowptr, cols, idx[ii], idx[jj]);
                if (p < 0) { err = ORC_ERR_PATTERN; continue; }
#pragma omp atomic
                vals[p] += K[jj + nsize * ii];
            }
        }
        for (ii = 0; ii < nsize; ++ii)
            if (idx[ii] == -1) {
                const int64_t n = conn[(int64_t)(ii / ndof) * nElem + e];
                const double fact = solnApplied[n * ndof + ii % ndof];
                for (jj = 0; jj < nsize; ++jj)
                    if (idx[jj] != -1) F[jj] = F[jj] - K[jj + nsize * ii] * fact;
            }
        for (ii = 0; ii < nsize; ++ii)
            if (idx[ii] >= 0) {
#pragma omp atomic
                rhs[idx[ii]] += F[ii];
            }
    }
    return err;
}

/* ------------------------------------------------------------------------ */
/* Linear algebra: CSR SpMV and Jacobi-PCG (PETSc KSPCG   */
/* semantics are third-party knowledge: PETSc 3.6-era, left-preconditioned,  */
/* KSP_NORM_PRECONDITIONED, zero initial guess per solverpetsc.F:459).       */
/* ------------------------------------------------------------------------ */

/* STREAM triad a = b + q*c on the host with the OpenMP threads of the baseline (first touch by the same threads): the
 * memory bandwidth the CPU figures of bench.py should be read against.  Test infrastructure, not a restatement. */
double orc_stream_triad_gbps(int64_t n, int reps)
{
    double *a = (double *)malloc(sizeof(double) * (size_t)n), *b = (double *)malloc(sizeof(double) * (size_t)n),
           *c = (double *)malloc(sizeof(double) * (size_t)n);
    double best = 0.0;
    int64_t i;
    int r;
    if (!a || !b || !c) { free(a); free(b); free(c); return 0.0; }
#pragma omp parallel for schedule(static)
    for (i = 0; i < n; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
    for (r = 0; r < reps; ++r) {
        struct timespec t0, t1;
        double dt;
        clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(static)
        for (i = 0; i < n; ++i) a[i] = b[i] + 3.0 * c[i];
        clock_gettime(CLOCK_MONOTONIC, &t1);
        dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (dt > 0.0 && 24.0e-9 * (double)n / dt > best) best = 24.0e-9 * (double)n / dt;
    }
    free(a); free(b); free(c);
    return best;
}

void orc_spmv(int64_t N, const int64_t *rowptr, const int32_t *cols, const double *vals,
              const double *x, double *y)
{
    int64_t r;
#pragma omp parallel for schedule(static) if (N > 200000)
    for (r = 0; r < N; ++r) {
        double s = 0.0;
        int64_t k;
        for (k = rowptr[r]; k < rowptr[r + 1]; ++k) s += vals[k] * x[cols[k]];
        y[r] = s;
    }
}

static double dotp(int64_t N, const double *a, const double *b)
{
    double s = 0.0;
    int64_t i;
#pragma omp parallel for reduction(+ : s) schedule(static) if (N > 200000)
    for (i = 0; i < N; ++i) s += a[i] * b[i];
    return s;
}

/* reason: 2 = converged rtol, 3 = converged atol, -3 = max its, -4 = dtol,
 * -8 = indefinite PC, -10 = indefinite matrix (PETSc KSPConvergedReason values). */
int orc_pcg_jacobi(int64_t N, const int64_t *rowptr, const int32_t *cols,
                   const double *vals, const double *b, double *x, double rtol,
                   double abstol, double dtol, int maxits, int *its_out,
                   int *reason_out, double *rnorm_out, double *history, int hist_len)
{
    double *r, *z, *p, *w, *dinv, beta, betan, rn0, rn, alpha, pw, ttol;
    int64_t i;
    int its = 0, reason = 0;
    r = (double *)malloc(sizeof(double) * 5 * (size_t)(N ? N : 1));
    if (!r) return ORC_ERR_NOMEM;
    z = r + N;  p = z + N;  w = p + N;  dinv = w + N;
    for (i = 0; i < N; ++i) {
        double d = 0.0;
        const int64_t k = csr_find(rowptr, cols, i, (int32_t)i);
        if (k >= 0) d = vals[k];
        dinv[i] = 1.0 / d;
    }
#pragma omp parallel for schedule(static) if (N > 200000)
    for (i = 0; i < N; ++i) { x[i] = 0.0;  r[i] = b[i];  z[i] = r[i] * dinv[i];  p[i] = z[i]; }
    beta = dotp(N, r, z);
    rn0 = sqrt(dotp(N, z, z));
    rn = rn0;
    if (history && hist_len > 0) history[0] = rn0;
    ttol = fmax(rtol * rn0, abstol);
    if (rn0 <= abstol) reason = 3;
    while (!reason) {
        if (its >= maxits) { reason = -3; break; }
        ++its;
        orc_spmv(N, rowptr, cols, vals, p, w);
        pw = dotp(N, p, w);
        if (!(pw > 0.0)) { reason = -10; break; }
        alpha = beta / pw;
#pragma omp parallel for schedule(static) if (N > 200000)
        for (i = 0; i < N; ++i) {
            x[i] += alpha * p[i];
            r[i] -= alpha * w[i];
            z[i] = r[i] * dinv[i];
        }
        betan = dotp(N, r, z);
        rn = sqrt(dotp(N, z, z));
        if (history && its < hist_len) history[its] = rn;
        if (rn <= ttol) { reason = rn <= abstol ? 3 : 2; break; }
        if (rn >= dtol * rn0) { reason = -4; break; }
        if (betan < 0.0) { reason = -8; break; }
        {
            const double bb = betan / beta;
#pragma omp parallel for schedule(static) if (N > 200000)
            for (i = 0; i < N; ++i) p[i] = z[i] + bb * p[i];
        }
        beta = betan;
    }
    *its_out = its;  *reason_out = reason;  *rnorm_out = rn;
    free(r);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* The same Jacobi-PCG in PETSc's single-reduction form (KSPCGUseSingleReduction, */
/* -ksp_cg_single_reduction; an option of the KSPCG that solverpetsc.F:187     */
/* creates, off by default).  Third-party semantics restated from the published */
/* algorithm (Chronopoulos & Gear 1989; PETSc is absent): S = A Z is formed     */
/* after every preconditioner application, (Z,S), (Z,R) and (Z,Z) are taken at */
/* one point, and                                                               */
/*     b = beta/beta_old,  dpi = delta - beta^2 dpi_old / beta_old^2,            */
/*     P = Z + b P,  W = S + b W,  a = beta/dpi,  X += a P,  R -= a W.           */
/* Same stopping rule, iteration numbering and reasons as orc_pcg_jacobi.        */
int orc_pcg_jacobi_single_reduction(int64_t N, const int64_t *rowptr, const int32_t *cols, const double *vals, const double *b,
                                    double *x, double rtol, double abstol, double dtol, int maxits, int *its_out,
                                    int *reason_out, double *rnorm_out, double *history, int hist_len)
{
    double *r, *z, *p, *w, *sv, *dinv, beta, beta_old = 0.0, delta, dpi = 0.0, rn0, rn, ttol;
    int64_t i;
    int its = 0, reason = 0;
    r = (double *)malloc(sizeof(double) * 6 * (size_t)(N ? N : 1));
    if (!r) return ORC_ERR_NOMEM;
    z = r + N;  p = z + N;  w = p + N;  sv = w + N;  dinv = sv + N;
    for (i = 0; i < N; ++i) {
        double d = 0.0;
        const int64_t k = csr_find(rowptr, cols, i, (int32_t)i);
        if (k >= 0) d = vals[k];
        dinv[i] = 1.0 / d;
    }
#pragma omp parallel for schedule(static) if (N > 200000)
    for (i = 0; i < N; ++i) { x[i] = 0.0;  r[i] = b[i];  z[i] = r[i] * dinv[i]; }
    orc_spmv(N, rowptr, cols, vals, z, sv);
    delta = dotp(N, z, sv);
    beta = dotp(N, r, z);
    rn0 = sqrt(dotp(N, z, z));
    rn = rn0;
    if (history && hist_len > 0) history[0] = rn0;
    ttol = fmax(rtol * rn0, abstol);
    if (rn0 <= abstol) reason = 3;
    else if (beta < 0.0) reason = -8;
    while (!reason) {
        double bb, a;
        if (its >= maxits) { reason = -3; break; }
        if (its == 0) { bb = 0.0;  dpi = delta; }
        else { bb = beta / beta_old;  dpi = delta - beta * beta * dpi / (beta_old * beta_old); }
        ++its;
        if (!(dpi > 0.0)) { reason = -10; break; }
        a = beta / dpi;
#pragma omp parallel for schedule(static) if (N > 200000)
        for (i = 0; i < N; ++i) {
            p[i] = its == 1 ? z[i] : z[i] + bb * p[i];
            w[i] = its == 1 ? sv[i] : sv[i] + bb * w[i];
            x[i] += a * p[i];
            r[i] -= a * w[i];
            z[i] = r[i] * dinv[i];
        }
        orc_spmv(N, rowptr, cols, vals, z, sv);
        beta_old = beta;
        delta = dotp(N, z, sv);
        beta = dotp(N, r, z);
        rn = sqrt(dotp(N, z, z));
        if (history && its < hist_len) history[its] = rn;
        if (rn <= ttol) { reason = rn <= abstol ? 3 : 2; break; }
        if (rn >= dtol * rn0) { reason = -4; break; }
        if (beta < 0.0) { reason = -8; break; }
    }
    *its_out = its;  *reason_out = reason;  *rnorm_out = rn;
    free(r);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* CG with the preconditioner the reference actually sets: PCBJACOBI, whose    */
/* default sub-solver is ILU(0) (solverpetsc.F:187, 206) -- one block per MPI */
/* rank, natural ordering.  block_start[0..nblocks] = row blocks (ascending);  */
/* entries coupling different blocks are dropped from the factorisation.       */
/* Third-party semantics restated from the published algorithm (PETSc is      */
/* absent): ILU(0) = Gaussian elimination on the pattern of A, no fill.        */
/* Same KSPCG loop and convergence test as orc_pcg_jacobi.                     */
/* ------------------------------------------------------------------------ */
int orc_pcg_bjacobi_ilu0(int64_t N, const int64_t *rowptr, const int32_t *cols, const double *vals, const double *b,
                         double *x, int nblocks, const int64_t *block_start, double rtol, double abstol, double dtol,
                         int maxits, int *its_out, int *reason_out, double *rnorm_out)
{
    double *lu, *r, *z, *p, *w, beta, betan, rn0, rn = 0.0, alpha, pw, ttol;
    int64_t *diag, i, k, kk, blk;
    int32_t *blk_of;
    int its = 0, reason = 0;
    lu = (double *)malloc(sizeof(double) * (size_t)(rowptr[N] ? rowptr[N] : 1));
    diag = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N ? N : 1));
    blk_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    r = (double *)malloc(sizeof(double) * 4 * (size_t)(N ? N : 1));
    if (!lu || !diag || !blk_of || !r) { free(lu); free(diag); free(blk_of); free(r); return ORC_ERR_NOMEM; }
    z = r + N;  p = z + N;  w = p + N;
    for (blk = 0; blk < nblocks; ++blk)
        for (i = block_start[blk]; i < block_start[blk + 1]; ++i) blk_of[i] = (int32_t)blk;
    memcpy(lu, vals, sizeof(double) * (size_t)rowptr[N]);
    for (i = 0; i < N; ++i) {
        diag[i] = csr_find(rowptr, cols, i, (int32_t)i);
        if (diag[i] < 0) { free(lu); free(diag); free(blk_of); free(r); return ORC_ERR_ARG; }
    }
    /* IKJ ILU(0) inside every diagonal block */
    for (i = 0; i < N; ++i) {
        const int64_t lo = block_start[blk_of[i]];
        for (k = rowptr[i]; k < diag[i]; ++k) {
            const int64_t c = cols[k];
            double f;
            if (c < lo) continue;                         /* another block: not part of this factor */
            f = lu[k] / lu[diag[c]];
            lu[k] = f;
            for (kk = diag[c] + 1; kk < rowptr[c + 1]; ++kk) {
                const int64_t j = csr_find(rowptr, cols, i, cols[kk]);
                if (j >= 0 && blk_of[cols[kk]] == blk_of[i]) lu[j] -= f * lu[kk];
            }
        }
    }
#define ORC_ILU_APPLY(rin, zout)                                                                     \
    for (i = 0; i < N; ++i) {                                                                        \
        const int64_t lo_ = block_start[blk_of[i]];                                                  \
        double s_ = (rin)[i];                                                                        \
        for (k = rowptr[i]; k < diag[i]; ++k) if (cols[k] >= lo_) s_ -= lu[k] * (zout)[cols[k]];     \
        (zout)[i] = s_;                                                                              \
    }                                                                                                \
    for (i = N - 1; i >= 0; --i) {                                                                   \
        const int64_t hi_ = block_start[blk_of[i] + 1];                                              \
        double s_ = (zout)[i];                                                                       \
        for (k = diag[i] + 1; k < rowptr[i + 1]; ++k) if (cols[k] < hi_) s_ -= lu[k] * (zout)[cols[k]]; \
        (zout)[i] = s_ / lu[diag[i]];                                                                \
    }
    for (i = 0; i < N; ++i) { x[i] = 0.0;  r[i] = b[i]; }
    ORC_ILU_APPLY(r, z)
    for (i = 0; i < N; ++i) p[i] = z[i];
    beta = dotp(N, r, z);
    rn0 = sqrt(dotp(N, z, z));
    rn = rn0;
    ttol = fmax(rtol * rn0, abstol);
    if (rn0 <= abstol) reason = 3;
    while (!reason) {
        if (its >= maxits) { reason = -3; break; }
        ++its;
        orc_spmv(N, rowptr, cols, vals, p, w);
        pw = dotp(N, p, w);
        if (!(pw > 0.0)) { reason = -10; break; }
        alpha = beta / pw;
        for (i = 0; i < N; ++i) { x[i] += alpha * p[i];  r[i] -= alpha * w[i]; }
        ORC_ILU_APPLY(r, z)
        betan = dotp(N, r, z);
        rn = sqrt(dotp(N, z, z));
        if (rn <= ttol) { reason = rn <= abstol ? 3 : 2; break; }
        if (rn >= dtol * rn0) { reason = -4; break; }
        if (betan < 0.0) { reason = -8; break; }
        { const double bb = betan / beta;  for (i = 0; i < N; ++i) p[i] = z[i] + bb * p[i]; }
        beta = betan;
    }
#undef ORC_ILU_APPLY
    *its_out = its;  *reason_out = reason;  *rnorm_out = rn;
    free(lu); free(diag); free(blk_of); free(r);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* Structured 6-tet box mesh (genTetra.cpp:152-216, 247-323, 348-525)        */
/* ------------------------------------------------------------------------ */

static double round8(double v)
{   /* the solver only ever sees the "%.8f" text (genTetra.cpp:187-189) */
    char buf[64];
    snprintf(buf, sizeof buf, "%.8f", v);
    return strtod(buf, NULL);
}

/* Fills xyz (SoA, as READ BACK from the %.8f file), conn (SoA, 0-based).
 * Dirichlet list: bc_mode 0 = all six faces with u = x^2+y^2+z^2 evaluated on
 * float-rounded coordinates (genTetra.cpp:510-525, vtkPoints is float);
 * bc_mode 1 = clamp plane y = y0 (all ndof dofs, value 0) -- the beam of
 * config 4; bc arrays may be NULL to only count (*nDBC).  */
int orc_gen_box_tets(double x0, double x1, int nEx, double y0, double y1, int nEy,
                     double z0, double z1, int nEz, int bc_mode, int ndof,
                     double *xyz, int32_t *conn, int64_t *nDBC, int32_t *bc_node,
                     int32_t *bc_dof, double *bc_val)
{
    const int nNx = nEx + 1, nNy = nEy + 1, nNz = nEz + 1;
    const int64_t nNode = (int64_t)nNx * nNy * nNz, nElem = 6LL * nEx * nEy * nEz;
    const double dx = (x1 - x0) / nEx, dy = (y1 - y0) / nEy, dz = (z1 - z0) / nEz;
    double *xs = (double *)malloc(sizeof(double) * (size_t)(nNx + nNy + nNz));
    double *ys = xs + nNx, *zs = ys + nNy, v;
    int64_t ind, e, nn = (int64_t)nNx * nNy, cnt = 0;
    int ii, jj, kk, d;
    if (!xs) return ORC_ERR_NOMEM;
    /* :194-216: xx starts at x0 for every row and is advanced by += dx */
    v = x0;  for (ii = 0; ii < nNx; ++ii) { xs[ii] = v;  v += dx; }
    v = y0;  for (jj = 0; jj < nNy; ++jj) { ys[jj] = v;  v += dy; }
    v = z0;  for (kk = 0; kk < nNz; ++kk) { zs[kk] = v;  v += dz; }
    if (xyz) {
        double *xr = (double *)malloc(sizeof(double) * (size_t)(nNx + nNy + nNz));
        double *yr = xr + nNx, *zr = yr + nNy;
        if (!xr) { free(xs); return ORC_ERR_NOMEM; }
        for (ii = 0; ii < nNx; ++ii) xr[ii] = round8(xs[ii]);
        for (jj = 0; jj < nNy; ++jj) yr[jj] = round8(ys[jj]);
        for (kk = 0; kk < nNz; ++kk) zr[kk] = round8(zs[kk]);
        ind = 0;
        for (kk = 0; kk < nNz; ++kk)
            for (jj = 0; jj < nNy; ++jj)
                for (ii = 0; ii < nNx; ++ii, ++ind) {
                    xyz[ind] = xr[ii];  xyz[nNode + ind] = yr[jj];  xyz[2 * nNode + ind] = zr[kk];
                }
        free(xr);
    }
    if (conn) {                                       /* :247-323 */
        e = 0;
        for (kk = 0; kk < nEz; ++kk)
            for (jj = 0; jj < nEy; ++jj)
                for (ii = 0; ii < nEx; ++ii) {
                    int32_t p[8];
                    static const int T[6][4] = {{0, 1, 3, 5}, {0, 3, 2, 5}, {2, 3, 7, 5},
                                                {4, 6, 7, 2}, {4, 7, 5, 2}, {0, 4, 5, 2}};
                    int t, a;
                    p[0] = (int32_t)(nn * kk + (int64_t)nNx * jj + ii);  p[1] = p[0] + 1;
                    p[2] = (int32_t)(nn * kk + (int64_t)nNx * (jj + 1) + ii);  p[3] = p[2] + 1;
                    p[4] = (int32_t)(nn * (kk + 1) + (int64_t)nNx * jj + ii);  p[5] = p[4] + 1;
                    p[6] = (int32_t)(nn * (kk + 1) + (int64_t)nNx * (jj + 1) + ii);  p[7] = p[6] + 1;
                    for (t = 0; t < 6; ++t, ++e)
                        for (a = 0; a < 4; ++a) conn[(int64_t)a * nElem + e] = p[T[t][a]];
                }
    }
    /* boundary nodes, ascending and unique (:505-506 sort+unique) */
    ind = 0;
    for (kk = 0; kk < nNz; ++kk)
        for (jj = 0; jj < nNy; ++jj)
            for (ii = 0; ii < nNx; ++ii, ++ind) {
                int on;
                if (bc_mode == 0)
                    on = ii == 0 || ii == nNx - 1 || jj == 0 || jj == nNy - 1 || kk == 0 || kk == nNz - 1;
                else
                    on = jj == 0;
                if (!on) continue;
                for (d = 0; d < ndof; ++d, ++cnt) {
                    if (!bc_node) continue;
                    bc_node[cnt] = (int32_t)ind;
                    bc_dof[cnt] = d;
                    if (bc_mode == 0) {
                        const double cx = (double)(float)xs[ii], cy = (double)(float)ys[jj],
                                     cz = (double)(float)zs[kk];
                        bc_val[cnt] = round8(cx * cx + cy * cy + cz * cz);   /* :518-524 */
                    } else
                        bc_val[cnt] = 0.0;
                }
            }
    *nDBC = cnt;
    free(xs);
    return ORC_OK;
}
