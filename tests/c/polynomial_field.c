/* go/r1csqaphip.PolynomialField.{Mul, Div, Add, Sub, Eval, LagrangeInterpolation, R1CSToQAP (Z), CombinePolynomials} as C
 * (go/gosnarkhip/seams.go: PolyAdd / PolySub / PolyEval / ZPoly / R1CSToPx; gosnarkhip.go: PolyMul / PolyDiv / LagrangeInterpolation).
 * The inputs are the reference's own test vectors (r1csqap/r1csqap_test.go:59-112) and the x^3 + x + 5 instance; every result is
 * written out and compared by the Python side with the oracle's restatement of r1csqap.go.
 * argv: r1cs file, groth16 instance (w, px), output. */
#include "instance.h"

int main(int argc, char** argv) {
  if (argc != 4) return 9;
  r1cs_instance q;
  groth_instance g;
  if (read_r1cs_instance(argv[1], &q) || read_groth_instance(argv[2], &g)) return 8;
  int dev = 0;
  CHECK(gs_init(&dev, 1));
  /* r1csqap_test.go:59-94: a = [1, 0, 5], b = [3, 0, 1] */
  const uint64_t a[12] = {1, 0, 0, 0, 0, 0, 0, 0, 5, 0, 0, 0}, b[12] = {3, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0};
  uint64_t out[4096];
  size_t pos = 0;
  CHECK(gs_poly_mul(a, 3, b, 3, out + pos)); pos += 5 * 4;                       /* Mul: 5 coefficients */
  CHECK(gs_poly_add(a, 3, b, 3, out + pos)); pos += 3 * 4;                       /* Add */
  CHECK(gs_poly_sub(a, 3, b, 3, out + pos)); pos += 3 * 4;                       /* Sub (mod r) */
  CHECK(gs_poly_div(out, 5, b, 3, out + pos, out + pos + 12)); pos += 3 * 4 + 2 * 4;   /* Div of the product by b: quotient a, remainder 0 */
  const uint64_t x[4] = {7, 0, 0, 0};
  CHECK(gs_poly_eval(a, 3, x, out + pos)); pos += 4;                              /* Eval(a, 7) = 1 + 5 * 49 */
  const uint64_t v[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5, 0, 0, 0};       /* r1csqap_test.go:107-112: LagrangeInterpolation([0,0,0,5]) */
  CHECK(gs_lagrange_interpolation(v, 4, out + pos)); pos += 4 * 4;
  CHECK(gs_zpoly(q.m - 2, out + pos)); pos += (q.m - 1) * 4;                      /* R1CSToQAP's Z: degree m - 2 */
  /* CombinePolynomials on the x^3 + x + 5 system: ax, bx, cx (n each), px (2n - 1) == the px the reference computed */
  uint64_t *ax = out + pos, *bx = ax + q.n * 4, *cx = bx + q.n * 4, *px = cx + q.n * 4;
  CHECK(gs_r1cs_to_px(q.n, q.m, q.rowptr[0], q.col[0], q.val[0], q.rowptr[1], q.col[1], q.val[1], q.rowptr[2], q.col[2], q.val[2], g.w, ax, bx, cx, px));
  pos += (3 * q.n + 2 * q.n - 1) * 4;
  if (g.npx != 2 * q.n - 1 || memcmp(px, g.px, g.npx * 32) != 0) { printf("FAIL: px differs from the reference's CombinePolynomials\n"); return 4; }
  if (write_words(argv[3], out, pos)) return 11;
  gs_shutdown();
  printf("OK\n");
  return 0;
}
