// cholmod_shim.cpp -- the seven CHOLMOD entry points Eigen's CholmodSupport calls (analyze / factorize / solve on a
// symmetric positive definite sparse matrix), implemented as a DENSE Cholesky behind the reference's own vendored
// <cholmod.h>.  SuiteSparse ships in the reference as headers + Windows DLLs only, so this is what lets the
// reference's FragmentOptimizer build and run here, unmodified (oracle/_ref/FragmentOptimizer_ref).
// TEST INFRASTRUCTURE: besides solving, every factorize / solve call can dump what the reference hands over --
// the assembled system matrix thisAA / thisJJ and the right-hand side -- which is how tests/test_fopt_oracle.py pins
// the restated assembly loops to the real ones:
//   ER_CHOLMOD_DUMP=<prefix>   ->  <prefix>_A<k>.bin : int64 n, int64 nnz, then nnz x (int32 row, int32 col, float64 value)
//                                   of the stored (upper) triangle handed to the k-th factorize call
//                                  <prefix>_b<k>.bin : int64 n, then n float64 (right-hand side of the k-th solve call)
#include <cholmod.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
int g_factorize_calls = 0, g_solve_calls = 0;

struct Dense {            // what we keep in cholmod_factor::x
  long n;
  std::vector<double> L;  // row-major lower Cholesky factor
  bool ok;
};

void dump_matrix(const cholmod_sparse* A) {
  const char* prefix = getenv("ER_CHOLMOD_DUMP");
  if (!prefix) return;
  const std::string fn = std::string(prefix) + "_A" + std::to_string(g_factorize_calls) + ".bin";
  FILE* f = fopen(fn.c_str(), "wb");
  if (!f) return;
  const int* Ap = static_cast<const int*>(A->p);
  const int* Ai = static_cast<const int*>(A->i);
  const int* Anz = static_cast<const int*>(A->nz);
  const double* Ax = static_cast<const double*>(A->x);
  int64_t n = (int64_t)A->nrow, nnz = 0;
  for (size_t j = 0; j < A->ncol; j++) {
    const int p0 = Ap[j], p1 = A->packed ? Ap[j + 1] : Ap[j] + Anz[j];
    for (int p = p0; p < p1; p++)
      if (A->stype == 0 || (A->stype > 0 && Ai[p] <= (int)j) || (A->stype < 0 && Ai[p] >= (int)j)) nnz++;
  }
  fwrite(&n, 8, 1, f);
  fwrite(&nnz, 8, 1, f);
  for (size_t j = 0; j < A->ncol; j++) {
    const int p0 = Ap[j], p1 = A->packed ? Ap[j + 1] : Ap[j] + Anz[j];
    for (int p = p0; p < p1; p++)
      if (A->stype == 0 || (A->stype > 0 && Ai[p] <= (int)j) || (A->stype < 0 && Ai[p] >= (int)j)) {
        int32_t r = Ai[p], c = (int32_t)j;
        fwrite(&r, 4, 1, f);
        fwrite(&c, 4, 1, f);
        fwrite(&Ax[p], 8, 1, f);
      }
  }
  fclose(f);
}
}  // namespace

extern "C" {

int cholmod_start(cholmod_common* c) {
  memset(c, 0, sizeof *c);
  c->status = CHOLMOD_OK;
  c->itype = CHOLMOD_INT;
  c->dtype = CHOLMOD_DOUBLE;
  return 1;
}
int cholmod_finish(cholmod_common*) { return 1; }

cholmod_factor* cholmod_analyze(cholmod_sparse* A, cholmod_common*) {
  cholmod_factor* L = static_cast<cholmod_factor*>(calloc(1, sizeof(cholmod_factor)));
  L->n = A->nrow;
  L->minor = A->nrow;
  Dense* d = new Dense();
  d->n = (long)A->nrow;
  d->ok = false;
  L->x = d;
  return L;
}

int cholmod_factorize(cholmod_sparse* A, cholmod_factor* Lf, cholmod_common* c) {
  dump_matrix(A);
  g_factorize_calls++;
  Dense* d = static_cast<Dense*>(Lf->x);
  const long n = d->n;
  d->L.assign((size_t)n * n, 0.0);
  const int* Ap = static_cast<const int*>(A->p);
  const int* Ai = static_cast<const int*>(A->i);
  const int* Anz = static_cast<const int*>(A->nz);
  const double* Ax = static_cast<const double*>(A->x);
  std::vector<double>& M = d->L;
  for (long j = 0; j < n; j++) {
    const int p0 = Ap[j], p1 = A->packed ? Ap[j + 1] : Ap[j] + Anz[j];
    for (int p = p0; p < p1; p++) {
      const long i = Ai[p];
      if (A->stype > 0 && i > j) continue;            // upper triangle stored: ignore anything below
      if (A->stype < 0 && i < j) continue;
      const long r = i > j ? i : j, q = i > j ? j : i;  // into the lower triangle
      M[(size_t)r * n + q] += Ax[p];
    }
  }
  // right-looking dense Cholesky, lower triangle, row-major
  d->ok = true;
  for (long k = 0; k < n; k++) {
    const double piv = M[(size_t)k * n + k];
    if (!(piv > 0.0)) {
      d->ok = false;
      Lf->minor = (size_t)k;
      if (c) c->status = CHOLMOD_NOT_POSDEF;
      return 1;
    }
    const double s = sqrt(piv);
    M[(size_t)k * n + k] = s;
    for (long i = k + 1; i < n; i++) M[(size_t)i * n + k] /= s;
    // 8 threads like the reference's own loops: hosts that expose hundreds of hardware threads under a CPU quota make wide
    // OpenMP barriers (one per column here) pathologically slow
#pragma omp parallel for schedule(dynamic, 16) num_threads(8) if (n - k > 256)
    for (long i = k + 1; i < n; i++) {
      const double lik = M[(size_t)i * n + k];
      if (lik == 0.0) continue;
      double* Mi = &M[(size_t)i * n];
      for (long j = k + 1; j <= i; j++) Mi[j] -= lik * M[(size_t)j * n + k];
    }
  }
  return 1;
}

cholmod_dense* cholmod_solve(int /*sys = CHOLMOD_A*/, cholmod_factor* Lf, cholmod_dense* B, cholmod_common*) {
  Dense* d = static_cast<Dense*>(Lf->x);
  const long n = d->n;
  if (const char* prefix = getenv("ER_CHOLMOD_DUMP")) {
    const std::string fn = std::string(prefix) + "_b" + std::to_string(g_solve_calls) + ".bin";
    if (FILE* f = fopen(fn.c_str(), "wb")) {
      int64_t nn = n;
      fwrite(&nn, 8, 1, f);
      fwrite(B->x, 8, (size_t)n, f);
      fclose(f);
    }
  }
  g_solve_calls++;
  cholmod_dense* X = static_cast<cholmod_dense*>(calloc(1, sizeof(cholmod_dense)));
  X->nrow = B->nrow;
  X->ncol = B->ncol;
  X->nzmax = B->nrow * B->ncol;
  X->d = B->nrow;
  X->xtype = B->xtype;
  X->dtype = B->dtype;
  double* x = static_cast<double*>(malloc(sizeof(double) * X->nzmax));
  X->x = x;
  const double* b = static_cast<const double*>(B->x);
  const std::vector<double>& M = d->L;
  if (!d->ok) {                                           // not positive definite (e.g. a run without correspondences): zeros, not NULL --
    memset(x, 0, sizeof(double) * X->nzmax);              // Eigen dereferences the result unconditionally
    return X;
  }
  for (size_t col = 0; col < B->ncol; col++) {
    double* y = x + col * X->d;
    const double* bc = b + col * B->d;
    for (long i = 0; i < n; i++) {                      // L y = b
      double s = bc[i];
      for (long j = 0; j < i; j++) s -= M[(size_t)i * n + j] * y[j];
      y[i] = s / M[(size_t)i * n + i];
    }
    for (long i = n - 1; i >= 0; i--) {                 // L^T x = y
      double s = y[i];
      for (long j = i + 1; j < n; j++) s -= M[(size_t)j * n + i] * y[j];
      y[i] = s / M[(size_t)i * n + i];
    }
  }
  return X;
}

int cholmod_free_dense(cholmod_dense** X, cholmod_common*) {
  if (X && *X) {
    free((*X)->x);
    free(*X);
    *X = nullptr;
  }
  return 1;
}

int cholmod_free_factor(cholmod_factor** L, cholmod_common*) {
  if (L && *L) {
    delete static_cast<Dense*>((*L)->x);
    free(*L);
    *L = nullptr;
  }
  return 1;
}

}  // extern "C"
