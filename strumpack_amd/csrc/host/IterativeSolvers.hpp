// strumpack::iterative::GMRes / BiCGStab: the two Krylov solvers the reference's structured-matrix examples precondition with a
// compressed matrix (iterative/IterativeSolvers.hpp:58-80; examples/dense/testStructured.cpp:139-185).  Host code on
// length-n vectors: the operator and the preconditioner are the caller's callbacks (the preconditioner is where the device
// works: StructuredMatrix::solve).  Own implementations of the textbook methods with the reference's argument lists:
// left-preconditioned restarted GMRES (Saad & Schultz; classical or modified Gram-Schmidt, Givens rotations) and BiCGStab
// (van der Vorst, the "templates" formulation).
#pragma once
#include <cmath>
#include <cstddef>
#include <functional>
#include <iomanip>
#include <iostream>
#include <vector>

#include "DenseMatrix.hpp"

namespace strumpack {

enum class GramSchmidtType { CLASSICAL, MODIFIED };

namespace iterative {

template <typename T> using SPMV = std::function<void(const T*, T*)>;
template <typename T> using PREC = std::function<void(T*)>;

namespace detail {
template <typename T> inline T dot(std::size_t n, const T* a, const T* b) { T s = 0; for (std::size_t i = 0; i < n; i++) s += a[i] * b[i]; return s; }
template <typename T> inline T nrm2(std::size_t n, const T* a) { return std::sqrt(dot(n, a, a)); }
}  // namespace detail

// solves M^{-1} A x = M^{-1} b; returns the last (preconditioned) residual norm; x and b: stride 1, length n
template <typename scalar_t, typename real_t = scalar_t>
real_t GMRes(const SPMV<scalar_t>& A, const PREC<scalar_t>& M, std::size_t n, scalar_t* x, const scalar_t* b, real_t rtol, real_t atol,
             int& totit, int maxit, int restart, GramSchmidtType GStype, bool non_zero_guess, bool verbose) {
  using detail::dot;
  using detail::nrm2;
  if (!non_zero_guess) for (std::size_t i = 0; i < n; i++) x[i] = scalar_t(0.);
  restart = std::max(1, std::min(restart, maxit > 0 ? maxit : 1));
  std::vector<std::vector<scalar_t>> V(restart + 1, std::vector<scalar_t>(n));
  std::vector<scalar_t> H((std::size_t)(restart + 1) * restart), cs(restart), sn(restart), g(restart + 1), w(n);
  real_t rho = 0, rho0 = 0;
  totit = 0;
  if (verbose) std::cout << "# GMRES: restart " << restart << ", " << (GStype == GramSchmidtType::CLASSICAL ? "classical" : "modified") << " Gram-Schmidt" << std::endl;
  bool done = false;
  while (!done && totit < maxit) {
    // r = M^{-1} (b - A x)
    if (non_zero_guess || totit > 0) { A(x, w.data()); for (std::size_t i = 0; i < n; i++) w[i] = b[i] - w[i]; }
    else for (std::size_t i = 0; i < n; i++) w[i] = b[i];
    M(w.data());
    rho = nrm2(n, w.data());
    if (totit == 0) rho0 = rho;
    if (rho == real_t(0.) || rho / rho0 < rtol || rho < atol) break;
    for (std::size_t i = 0; i < n; i++) V[0][i] = w[i] / rho;
    std::fill(g.begin(), g.end(), scalar_t(0.));
    g[0] = rho;
    int k = 0;
    for (; k < restart && totit < maxit; k++) {
      A(V[k].data(), w.data());
      M(w.data());
      scalar_t* h = &H[(std::size_t)k * (restart + 1)];
      if (GStype == GramSchmidtType::CLASSICAL) {
        for (int j = 0; j <= k; j++) h[j] = dot(n, V[j].data(), w.data());
        for (int j = 0; j <= k; j++) for (std::size_t i = 0; i < n; i++) w[i] -= h[j] * V[j][i];
      } else {
        for (int j = 0; j <= k; j++) { h[j] = dot(n, V[j].data(), w.data()); for (std::size_t i = 0; i < n; i++) w[i] -= h[j] * V[j][i]; }
      }
      h[k + 1] = nrm2(n, w.data());
      if (h[k + 1] != scalar_t(0.)) for (std::size_t i = 0; i < n; i++) V[k + 1][i] = w[i] / h[k + 1];
      for (int j = 0; j < k; j++) { const scalar_t t = cs[j] * h[j] + sn[j] * h[j + 1]; h[j + 1] = -sn[j] * h[j] + cs[j] * h[j + 1]; h[j] = t; }
      const scalar_t d = std::hypot(h[k], h[k + 1]);
      cs[k] = d != scalar_t(0.) ? h[k] / d : scalar_t(1.);
      sn[k] = d != scalar_t(0.) ? h[k + 1] / d : scalar_t(0.);
      h[k] = d;
      h[k + 1] = 0;
      g[k + 1] = -sn[k] * g[k];
      g[k] = cs[k] * g[k];
      rho = std::abs(g[k + 1]);
      totit++;
      if (verbose) std::cout << "GMRES it. " << totit << "\tres = " << std::setw(12) << rho << "\trel.res = " << std::setw(12) << rho / rho0 << std::endl;
      if (rho / rho0 < rtol || rho < atol) { done = true; k++; break; }
    }
    // y = H(0:k, 0:k)^{-1} g(0:k);  x += V(:, 0:k) y
    std::vector<scalar_t> y(k);
    for (int i = k - 1; i >= 0; i--) {
      scalar_t s = g[i];
      for (int j = i + 1; j < k; j++) s -= H[(std::size_t)j * (restart + 1) + i] * y[j];
      y[i] = s / H[(std::size_t)i * (restart + 1) + i];
    }
    for (int j = 0; j < k; j++) for (std::size_t i = 0; i < n; i++) x[i] += y[j] * V[j][i];
  }
  return rho;
}

// solves A x = b with the (right) preconditioner M; returns the last residual norm
template <typename scalar_t, typename real_t = scalar_t>
real_t BiCGStab(const SPMV<scalar_t>& A, const PREC<scalar_t>& M, std::size_t n, scalar_t* x, const scalar_t* b, real_t rtol, real_t atol,
                int& totit, int maxit, bool non_zero_guess, bool verbose) {
  using detail::dot;
  using detail::nrm2;
  std::vector<scalar_t> r(n), rt(n), p(n), v(n), s(n), t(n), ph(n), sh(n);
  if (!non_zero_guess) for (std::size_t i = 0; i < n; i++) x[i] = scalar_t(0.);
  A(x, r.data());
  for (std::size_t i = 0; i < n; i++) r[i] = b[i] - r[i];
  rt = r;
  real_t bnrm = nrm2(n, b);
  if (bnrm == real_t(0.)) bnrm = 1;
  real_t res = nrm2(n, r.data());
  scalar_t rho = 1, rho1 = 1, alpha = 1, omega = 1;
  totit = 0;
  if (verbose) std::cout << "# BiCGStab" << std::endl;
  while (totit < maxit && !(res / bnrm < rtol || res < atol)) {
    rho = dot(n, rt.data(), r.data());
    if (rho == scalar_t(0.)) break;
    if (totit == 0) p = r;
    else {
      const scalar_t beta = (rho / rho1) * (alpha / omega);
      for (std::size_t i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - omega * v[i]);
    }
    ph = p;
    M(ph.data());
    A(ph.data(), v.data());
    alpha = rho / dot(n, rt.data(), v.data());
    for (std::size_t i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
    totit++;
    if (nrm2(n, s.data()) < atol) {
      for (std::size_t i = 0; i < n; i++) x[i] += alpha * ph[i];
      res = nrm2(n, s.data());
      break;
    }
    sh = s;
    M(sh.data());
    A(sh.data(), t.data());
    omega = dot(n, t.data(), s.data()) / dot(n, t.data(), t.data());
    for (std::size_t i = 0; i < n; i++) { x[i] += alpha * ph[i] + omega * sh[i]; r[i] = s[i] - omega * t[i]; }
    res = nrm2(n, r.data());
    if (verbose) std::cout << "BiCGStab it. " << totit << "\tres = " << std::setw(12) << res << "\trel.res = " << std::setw(12) << res / bnrm << std::endl;
    if (omega == scalar_t(0.)) break;
    rho1 = rho;
  }
  return res;
}

}  // namespace iterative
}  // namespace strumpack
