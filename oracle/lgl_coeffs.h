/* ORACLE (test infrastructure, never shipped or imported by the product path).
 *
 * LGL3 / LGL5 / LGL7 collocation coefficient tables, restated from
 * /root/reference/src/OptimalControl/LGLCoeffs.h:15-56 (CS=2), :62-162 (CS=3), :169-393 (CS=4).
 * Only the tables the defect path uses are kept.  The numeric literals (and, for LGL5, the
 * closed-form expressions in SQRT21) are reproduced digit-for-digit on purpose: SURVEY.md
 * appendix B shows that re-deriving "exact" coefficients moves residuals at the 1e-15 level.
 *
 * Naming used across this build ("Mathematical statement"):
 *   s = InteriorSpacings            A = Cardinal_XInterp_Weights   B = Cardinal_DXInterp_Weights
 *   U = Cardinal_UPoly_Weights      C = Cardinal_XDef_Weights      D = Cardinal_DXDef_Weights
 *   E = Interior_DXDef_Weights      tc = CardinalSpacings
 */
#ifndef ORACLE_LGL_COEFFS_H
#define ORACLE_LGL_COEFFS_H

typedef struct {
  int cs;            /* cardinal nodes per segment */
  double tc[4];      /* CardinalSpacings  [cs]   */
  double s[3];       /* InteriorSpacings  [cs-1] */
  double A[3][4];    /* [interior][cardinal] */
  double B[3][4];
  double U[3][4];
  double C[3][4];
  double D[3][4];
  double E[3];
} lgl_tables;

#define LGL_SQRT21 4.58257569495584
#define LGL_R686 (1.0 / 686.0)
#define LGL_R360A (1.0 / 360)
#define LGL_R360B (1.0 / 360.0)

static const lgl_tables LGL_TABLES[3] = {
    /* ---------------- CS = 2 : LGL3, cubic Hermite ---------------- */
    {2,
     {0.0, 1.0},
     {0.5},
     {{0.5, 0.5}},
     {{0.125, -0.125}},
     {{0.5, 0.5}},
     {{1.0, -1.0}},
     {{1.0 / 6.0, 1.0 / 6.0}},
     {4.0 / 6.0}},
    /* ---------------- CS = 3 : LGL5 ---------------- */
    {3,
     {0.0, 0.5, 1.0},
     {0.172673164646011, 0.827326835353989},
     {{(39.0 * LGL_SQRT21 + 231.0) * LGL_R686, 224.0 * LGL_R686, (-39.0 * LGL_SQRT21 + 231.0) * LGL_R686},
      {(-39.0 * LGL_SQRT21 + 231.0) * LGL_R686, 224.0 * LGL_R686, (39.0 * LGL_SQRT21 + 231.0) * LGL_R686}},
     {{(3.0 * LGL_SQRT21 + 21.0) * LGL_R686, (-16.0 * LGL_SQRT21) * LGL_R686, (3.0 * LGL_SQRT21 - 21.0) * LGL_R686},
      {(-3.0 * LGL_SQRT21 + 21.0) * LGL_R686, (16.0 * LGL_SQRT21) * LGL_R686, (-3.0 * LGL_SQRT21 - 21.0) * LGL_R686}},
     {{0.541612549639704, 0.571428571428571, -0.113041121068274},
      {-0.113041121068274, 0.571428571428571, 0.541612549639704}},
     {{(32.0 * LGL_SQRT21 + 180.0) * LGL_R360A, -64.0 * LGL_SQRT21 * LGL_R360A, (32.0 * LGL_SQRT21 - 180.0) * LGL_R360A},
      {(-32.0 * LGL_SQRT21 + 180.0) * LGL_R360B, 64.0 * LGL_SQRT21 * LGL_R360B, (-32.0 * LGL_SQRT21 - 180.0) * LGL_R360B}},
     {{(9.0 + LGL_SQRT21) * LGL_R360A, 64.0 * LGL_R360A, (9.0 - LGL_SQRT21) * LGL_R360A},
      {(9.0 - LGL_SQRT21) * LGL_R360B, 64.0 * LGL_R360B, (9.0 + LGL_SQRT21) * LGL_R360B}},
     {98.0 * LGL_R360A, 98.0 * LGL_R360B}},
    /* ---------------- CS = 4 : LGL7 ---------------- */
    {4,
     {+0.00000000000000, +2.65575603264643e-1, +7.34424396735357e-1, +1.00000000000000},
     {+8.48880518607166e-2, +0.50000000000000, +9.15111948139283e-1},
     {{+6.18612232711785e-1, +3.34253095933642e-1, +1.52679626438851e-2, +3.18667087106879e-2},
      {+1.41445282326366e-1, +3.58554717673634e-1, +3.58554717673634e-1, +1.41445282326366e-1},
      {+3.18667087106879e-2, +1.52679626438851e-2, +3.34253095933642e-1, +6.18612232711785e-1}},
     {{+2.57387738427162e-2, -5.50098654524528e-2, -1.53026046503702e-2, -2.38759243962924e-3},
      {+9.92317607754556e-3, +9.62835932121973e-2, -9.62835932121973e-2, -9.92317607754556e-3},
      {+2.38759243962924e-3, +1.53026046503702e-2, +5.50098654524528e-2, -2.57387738427162e-2}},
     {{0.550643660407289, 0.551767574740443, -0.153490305524281, 0.0510790703765507},
      {-0.140877081724073, 0.640877081724073, 0.640877081724073, -0.140877081724073},
      {0.0510790703765507, -0.153490305524281, 0.551767574740443, 0.550643660407289}},
     {{+8.84260109348311e-1, -8.23622559094327e-1, -2.35465327970606e-2, -3.70910174569208e-2},
      {+7.86488731947674e-2, +8.00076026297266e-1, -8.00076026297266e-1, -7.86488731947674e-2},
      {+3.70910174569208e-2, +2.35465327970606e-2, +8.23622559094327e-1, -8.84260109348311e-1}},
     {{+1.62213410652341e-2, +9.71662045547156e-2, +1.85682012187242e-2, +2.74945307600086e-3},
      {+4.83872966828888e-3, +1.00138284831491e-1, +1.00138284831491e-1, +4.83872966828888e-3},
      {+2.74945307600086e-3, +1.85682012187242e-2, +9.71662045547156e-2, +1.62213410652341e-2}},
     {+1.38413023680783e-1, +2.43809523809524e-1, +1.38413023680783e-1}},
};

static inline const lgl_tables* lgl_get(int cs) {
  return (cs >= 2 && cs <= 4) ? &LGL_TABLES[cs - 2] : (const lgl_tables*)0;
}


/* Quadrature and control-spline weights used by the other per-segment functions of a phase (oracle/pathfuncs.cpp),
 * restated from LGLCoeffs.h: Reduced_Integral_Weights :42, :135, :360-364; UZeroSpline_Weights / UOneSpline_Weights
 * :155-158 (CS=3), :372-388 (CS=4).  The literals and the expressions are the reference's. */
static const double lgl_reduced_integral_weights_[3][4] = {
    {0.5, 0.5, 0.0, 0.0},
    {1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0, 0.0},
    {-5.12701665379258 / 4.0 + 10.2540333075852 / 3.0 - 6.12701665379258 / 2.0 + 1.0,
     10.9353308042859 / 4.0 - 18.9665045333251 / 3.0 + 8.03117372903925 / 2.0,
     -10.9353308042859 / 4.0 + 13.8394878795326 / 3.0 - 2.90415707524666 / 2.0,
     5.12701665379258 / 4.0 - 5.12701665379258 / 3.0 + 1.0 / 2.0}};
static const double lgl_uzero_spline_weights_[2][2][4] = {
    {{-3.0, 4.0, -1.0, 0.0}, {0.0, 0.0, 0.0, 0.0}},
    {{-6.12701665379258, 8.03117372903925, -2.90415707524666, 1.0},
     {10.2540333075852 * 2.0, -18.9665045333251 * 2.0, +13.8394878795326 * 2.0, -5.12701665379258 * 2.0}}};
static const double lgl_uone_spline_weights_[2][2][4] = {
    {{1.0, -4.0, 3.0, 0.0}, {0.0, 0.0, 0.0, 0.0}},
    {{-5.12701665379258 * 3.0 + 10.2540333075852 * 2.0 - 6.12701665379258,
      10.9353308042859 * 3.0 - 18.9665045333251 * 2.0 + 8.03117372903925,
      -10.9353308042859 * 3.0 + 13.8394878795326 * 2.0 - 2.90415707524666,
      5.12701665379258 * 3.0 - 5.12701665379258 * 2.0 + 1.0},
     {-5.12701665379258 * 6.0 + 10.2540333075852 * 2.0, 10.9353308042859 * 6.0 - 18.9665045333251 * 2.0,
      -10.9353308042859 * 6.0 + 13.8394878795326 * 2.0, 5.12701665379258 * 6.0 - 5.12701665379258 * 2.0}}};
static inline const double* oracle_reduced_integral_weights(int cs) {
  return (cs >= 2 && cs <= 4) ? lgl_reduced_integral_weights_[cs - 2] : 0;
}
static inline const double (*oracle_uzero_spline_weights(int cs))[4] {
  return (cs == 3 || cs == 4) ? lgl_uzero_spline_weights_[cs - 3] : 0;
}
static inline const double (*oracle_uone_spline_weights(int cs))[4] {
  return (cs == 3 || cs == 4) ? lgl_uone_spline_weights_[cs - 3] : 0;
}

#endif
