/* Test-only driver: the CPU oracle (oracle/oi_oracle.c) compiled with AddressSanitizer + UBSan and run over a few small
 * ragged volumes: mask, assembled rows + checkMatrixProperties, codes, CG, fluxes, tau, homogenization rows. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef long long i64;
void oi_oracle_gen_spheres(uint8_t*, int, int, int, uint64_t, int, int);
void oi_oracle_gen_bernoulli(uint8_t*, int, int, int, uint64_t, uint64_t);
i64 oi_oracle_active_mask(const uint8_t*, int, int, int, int, int, uint8_t*);
void oi_oracle_fillmtx(const uint8_t*, const uint8_t*, int, int, int, int, int, const double*, double, double, double*, double*, double*);
i64 oi_oracle_check_matrix(const double*, const double*, const uint8_t*, int, int, int, int, double, double);
void oi_oracle_codes(const uint8_t*, int, int, int, int, uint8_t*);
void oi_oracle_initial_field(const uint8_t*, int, int, int, int, double, double, double*);
double oi_oracle_bnorm(const uint8_t*, int, int, int, int, double, double);
int oi_oracle_cg(const uint8_t*, int, int, int, double, double, int, int, double*, int*, double*);
void oi_oracle_fluxes(const double*, const uint8_t*, int, int, int, int, double*, double*, i64*, i64*);
double oi_oracle_tau(double, int, double, double, int, int, int, int, double, double, double*, int*);
void oi_oracle_remspot(uint8_t*, int, int, int);

int main(void) {
  const int shapes[3][3] = {{17, 9, 12}, {8, 8, 8}, {5, 31, 6}};
  for (int s = 0; s < 3; ++s) {
    const int nx = shapes[s][0], ny = shapes[s][1], nz = shapes[s][2];
    const size_t n = (size_t)nx * ny * nz;
    uint8_t* ph = malloc(n), *mask = malloc(n), *code = malloc(n);
    double* x = malloc(n * 8), *a = malloc(n * 7 * 8), *rhs = malloc(n * 8), *xi = malloc(n * 8);
    oi_oracle_gen_bernoulli(ph, nx, ny, nz, 7 + s, (uint64_t)(0.7 * 9007199254740992.0));
    oi_oracle_remspot(ph, nx, ny, nz);
    for (int dir = 0; dir < 3; ++dir) {
      const i64 nact = oi_oracle_active_mask(ph, nx, ny, nz, 1, dir, mask);
      const double dxinv[3] = {1.0, 1.0, 1.0};
      oi_oracle_fillmtx(ph, mask, nx, ny, nz, 1, dir, dxinv, -1.0, 2.0, a, rhs, xi);
      const i64 viol = oi_oracle_check_matrix(a, rhs, mask, nx, ny, nz, dir, -1.0, 2.0);
      oi_oracle_codes(mask, nx, ny, nz, dir, code);
      oi_oracle_initial_field(mask, nx, ny, nz, dir, -1.0, 2.0, x);
      int it = 0, st = 0;
      double rel = 0, fi = 0, fo = 0, deff = 0;
      i64 ni = 0, no = 0;
      const int conv = nact ? oi_oracle_cg(code, nx, ny, nz, oi_oracle_bnorm(mask, nx, ny, nz, dir, -1.0, 2.0), 1e-10, 5000, 1, x, &it, &rel) : 0;
      oi_oracle_fluxes(x, mask, nx, ny, nz, dir, &fi, &fo, &ni, &no);
      const double tau = oi_oracle_tau((double)nact / (double)n, conv, fi, fo, nx, ny, nz, dir, -1.0, 2.0, &deff, &st);
      printf("shape %d dir %d: active %lld violations %lld iterations %d tau %.6f status %d\n", s, dir, nact, viol, it, tau, st);
      if (viol) return 1;
    }
    free(ph); free(mask); free(code); free(x); free(a); free(rhs); free(xi);
  }
  printf("oracle done\n");
  return 0;
}
