// HSSMatrix<float>, HSSMatrix<std::complex<float>>, HSSMatrix<std::complex<double>>: the other three instantiations
// of the reference (HSS/HSSMatrix.cpp:513-516) behind the same construct / mult / factor / solve surface, computed by
// the double-precision device engine:
//   float            -> promoted to double (the FP64 engine is the hot path of this library; results are at least as
//                       accurate as a single-precision computation, the memory footprint is that of the double matrix);
//   complex<double>  -> the real image of A = Ar + i Ai with real and imaginary parts INTERLEAVED,
//                         Ahat(2i+a, 2j+b) = [Ar_ij  -Ai_ij ; Ai_ij  Ar_ij](a, b),
//                       a real 2n x 2n matrix with the same HSS structure (cluster sizes and ranks doubled).  With this
//                       ordering a complex vector IS its real image in memory (re, im, re, im ...), so mult / solve
//                       operate on the caller's complex buffers in place; Ahat^T is the image of A^H.  Column 2j of
//                       Ahat is column j of A viewed as reals, column 2j+1 the image of i A(:, j): the operand is
//                       streamed through the device block by block IN ITS OWN FORMAT (half the bytes of the image, a
//                       quarter for complex<float>; a float matrix half the bytes of its promotion) and expanded there
//                       (hssk_expand_image, DeviceHSS::HostBlockSource), never stored;
//   complex<float>   -> promoted to complex<double>.
// Cost of the embedding against a native complex engine: 2x the stored reals and ~2x the flops at equal accuracy
// (4 real multiplications per complex one either way, but the sketch carries 2r instead of r columns).
// rank() reports complex ranks (half the embedded rank, rounded up).
#pragma once
#include <complex>
#include <memory>
#include <type_traits>
#include <vector>

#include "HSSMatrix.hpp"

namespace strumpack {
namespace HSS {

template <typename T> struct promoted_traits { static constexpr bool cplx = false; using real_t = T; };
template <typename R> struct promoted_traits<std::complex<R>> { static constexpr bool cplx = true; using real_t = R; };

template <typename T> class HSSMatrixPromoted : public structured::StructuredMatrix<T> {
 public:
  using scalar_t = T;
  using real_t = typename promoted_traits<T>::real_t;
  using DenseM_t = DenseMatrix<T>;
  using opts_t = HSSOptions<double>;
  using elem_t = std::function<void(const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseM_t& B)>;
  static constexpr bool cplx = promoted_traits<T>::cplx;
  static constexpr std::size_t W = cplx ? 2 : 1;   // reals per scalar

  HSSMatrixPromoted() {}
  HSSMatrixPromoted(const DenseM_t& A, const opts_t& opts) : HSSMatrixPromoted(A.rows(), A.cols(), opts) { compress(A, opts); }
  HSSMatrixPromoted(std::size_t m, std::size_t n, const opts_t& opts) : rows_(m), cols_(n) {
    if (m != n) throw std::invalid_argument("HSS compression only supported for square matrices.");
    make(opts, nullptr);
  }
  HSSMatrixPromoted(const structured::ClusterTree& t, const opts_t& opts) : rows_(t.size), cols_(t.size) { make(opts, &t); }

  void compress(const DenseM_t& A, const opts_t& opts) {
    if (A.rows() != rows_ || A.cols() != cols_) throw std::invalid_argument("compress: matrix dimensions do not match");
    // the operand crosses the link in its own format and is expanded to its image on the device
    const int dtype = std::is_same<T, float>::value ? 1 : std::is_same<T, std::complex<float>>::value ? 2 : 3;   // HSSK_DT_*
    H_->compress_image(A.data(), A.ld(), dtype, scaled(opts));
  }
  void compress(const elem_t& Aelem, const opts_t& opts) {
    // block evaluation of the image from block evaluations of A
    typename HSSMatrix<double>::elem_t img = [&](const std::vector<std::size_t>& I, const std::vector<std::size_t>& J, DenseMatrix<double>& B) {
      // scalar rows / columns behind the requested real ones (the two reals of one complex entry are usually adjacent in
      // the request: evaluated once)
      std::vector<std::size_t> Is, Js, ri(I.size()), cj(J.size());
      for (std::size_t i = 0; i < I.size(); i++) {
        if (Is.empty() || Is.back() != I[i] / W) Is.push_back(I[i] / W);
        ri[i] = Is.size() - 1;
      }
      for (std::size_t j = 0; j < J.size(); j++) {
        if (Js.empty() || Js.back() != J[j] / W) Js.push_back(J[j] / W);
        cj[j] = Js.size() - 1;
      }
      DenseM_t Bt(Is.size(), Js.size());
      Aelem(Is, Js, Bt);
      for (std::size_t j = 0; j < J.size(); j++)
        for (std::size_t i = 0; i < I.size(); i++) B(i, j) = image(Bt(ri[i], cj[j]), I[i] % W, J[j] % W);
    };
    H_->compress_from_elements(img, scaled(opts));
  }

  std::size_t rows() const override { return rows_; }
  std::size_t cols() const override { return cols_; }
  std::size_t memory() const override { return H_ ? H_->memory() : 0; }
  std::size_t nonzeros() const override { return H_ ? H_->nonzeros() / W : 0; }
  std::size_t rank() const override { return H_ ? (H_->rank() + W - 1) / W : 0; }
  std::size_t levels() const { return H_ ? H_->levels() : 0; }
  bool is_compressed() const { return H_ && H_->is_compressed(); }

  void mult(Trans op, const DenseM_t& x, DenseM_t& y) const override {
    if (x.cols() != y.cols() || x.rows() != rows_ || y.rows() != rows_) throw std::invalid_argument("mult: dimensions do not match");
    mult(op, int(x.cols()), x.data(), x.ld(), y.data(), y.ld());
  }
  // y = op(A) x on raw column-major buffers (the C interface's form)
  void mult(Trans op, int nrhs, const T* x, int ldx, T* y, int ldy) const {
    const std::size_t n = rows_;
    if (std::is_same<T, std::complex<double>>::value && op != Trans::T) {
      // the complex buffers ARE the real images: no copy.  A -> Ahat, A^H -> Ahat^T
      DenseMatrixWrapper<double> X(2 * n, nrhs, reinterpret_cast<double*>(const_cast<T*>(x)), 2 * std::size_t(ldx));
      DenseMatrixWrapper<double> Y(2 * n, nrhs, reinterpret_cast<double*>(y), 2 * std::size_t(ldy));
      H_->mult(op == Trans::N ? Trans::N : Trans::T, X, Y);
      return;
    }
    // float / complex<float>: promote; plain transpose of a complex matrix: A^T x = conj(A^H conj(x))
    const bool conj = cplx && op == Trans::T;
    DenseMatrix<double> X(W * n, nrhs), Y(W * n, nrhs);
    for (int c = 0; c < nrhs; c++)
      for (std::size_t i = 0; i < n; i++) put(X, i, c, x[i + std::size_t(ldx) * c], conj);
    H_->mult(op == Trans::N ? Trans::N : Trans::T, X, Y);
    for (int c = 0; c < nrhs; c++)
      for (std::size_t i = 0; i < n; i++) y[i + std::size_t(ldy) * c] = get(Y, i, c, conj);
  }
  void factor() override { H_->factor(); }
  void solve(DenseM_t& b) const override { solve(int(b.cols()), b.data(), b.ld()); }
  void solve(int nrhs, T* b, int ldb) const override {
    const std::size_t n = rows_;
    if (std::is_same<T, std::complex<double>>::value) {
      DenseMatrixWrapper<double> B(2 * n, nrhs, reinterpret_cast<double*>(b), 2 * std::size_t(ldb));
      H_->solve(B);
      return;
    }
    DenseMatrix<double> B(W * n, nrhs);
    for (int c = 0; c < nrhs; c++)
      for (std::size_t i = 0; i < n; i++) put(B, i, c, b[i + std::size_t(ldb) * c], false);
    H_->solve(B);
    for (int c = 0; c < nrhs; c++)
      for (std::size_t i = 0; i < n; i++) b[i + std::size_t(ldb) * c] = get(B, i, c, false);
  }
  void shift(T s) override {
    if (cplx) H_->engine()->shift_cplx(re(s), im(s));
    else H_->shift(re(s));
  }
  // the double-precision matrix that carries the computation (n x n, or the 2n x 2n real image)
  const HSSMatrix<double>& carrier() const { return *H_; }

 private:
  static double re(const T& v) { return double(std::real(v)); }
  static double im(const T& v) { return double(std::imag(v)); }
  // entry (a, b) of the 2 x 2 image [re -im; im re] of v (real types: v itself)
  static double image(const T& v, std::size_t a, std::size_t b) {
    if (!cplx) return re(v);
    return a == b ? re(v) : (a ? im(v) : -im(v));
  }
  static void put(DenseMatrix<double>& X, std::size_t i, int c, const T& v, bool conj) {
    if (!cplx) { X(i, c) = re(v); return; }
    X(2 * i, c) = re(v);
    X(2 * i + 1, c) = conj ? -im(v) : im(v);
  }
  static T make_scalar(double r, double i, std::true_type) { return T(real_t(r), real_t(i)); }
  static T make_scalar(double r, double, std::false_type) { return T(r); }
  static T get(const DenseMatrix<double>& X, std::size_t i, int c, bool conj) {
    if (!cplx) return make_scalar(X(i, c), 0., std::integral_constant<bool, cplx>());
    return make_scalar(X(2 * i, c), conj ? -X(2 * i + 1, c) : X(2 * i + 1, c), std::integral_constant<bool, cplx>());
  }
  // the image's options: cluster sizes, ranks and sample counts are counted in reals
  opts_t scaled(const opts_t& o) const {
    opts_t s(o);
    if (cplx) {
      s.set_leaf_size(2 * o.leaf_size());
      s.set_d0(2 * o.d0());
      s.set_dd(2 * o.dd());
      s.set_p(2 * o.p());
      if (o.max_rank() < (1 << 29)) s.set_max_rank(2 * o.max_rank());
    }
    return s;
  }
  // cluster tree of the image: the scalar tree (given, or the reference's bisection, HSS/HSSMatrix.cpp:60-70) with every
  // size doubled, so that no complex entry is split between two clusters
  static structured::ClusterTree doubled(const structured::ClusterTree& t) {
    structured::ClusterTree d(int(W) * t.size);
    d.c.reserve(t.c.size());
    for (auto& c : t.c) d.c.push_back(doubled(c));
    return d;
  }
  static structured::ClusterTree bisect(int m, int leaf) {
    structured::ClusterTree t(m);
    if (m > leaf) { t.c.push_back(bisect(m / 2, leaf)); t.c.push_back(bisect(m - m / 2, leaf)); }
    return t;
  }
  void make(const opts_t& o, const structured::ClusterTree* t) {
    structured::ClusterTree base = t ? *t : bisect(int(rows_), o.leaf_size());
    H_.reset(new HSSMatrix<double>(doubled(base), scaled(o)));
  }
  std::size_t rows_ = 0, cols_ = 0;
  std::unique_ptr<HSSMatrix<double>> H_;
};

template <> class HSSMatrix<float> : public HSSMatrixPromoted<float> { public: using HSSMatrixPromoted<float>::HSSMatrixPromoted; };
template <> class HSSMatrix<std::complex<float>> : public HSSMatrixPromoted<std::complex<float>> { public: using HSSMatrixPromoted<std::complex<float>>::HSSMatrixPromoted; };
template <> class HSSMatrix<std::complex<double>> : public HSSMatrixPromoted<std::complex<double>> { public: using HSSMatrixPromoted<std::complex<double>>::HSSMatrixPromoted; };

}  // namespace HSS
}  // namespace strumpack
