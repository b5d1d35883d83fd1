// The reference's default random stream -- std::minstd_rand seeded with 0 under std::normal_distribution<double>
// (misc/RandomWrapper.hpp:128-191, libstdc++) -- reproduced bit for bit and generated on all host threads.
//
// The serial form costs 20-35 ns per number: 0.4-0.65 s for the 100000 x 192 sketching matrix of one compression round,
// several times the device time of the whole compression.  It parallelises exactly:
//   * libstdc++'s normal_distribution is the polar (Marsaglia) method: an ATTEMPT draws x and y uniformly in (-1, 1) and
//     is accepted when 0 < x^2 + y^2 <= 1; an accepted attempt yields y * mult, then x * mult (the saved value);
//   * each uniform is generate_canonical<double, 53>, which takes exactly two engine draws for minstd_rand (range 2^31 - 2),
//     so every attempt consumes FOUR draws whatever its outcome: attempt j starts at draw 4 j of the engine's stream;
//   * minstd_rand is the multiplicative generator x <- 48271 x mod (2^31 - 1): draw k is a^k x0 mod m, a jump of any
//     length is one modular power.
// So a range of attempts can be evaluated from nothing but its first index.  Ranges go to the host threads, the accepted
// pairs are concatenated in attempt order, and the engine is left exactly where the serial loop would have left it
// (including the saved second value of the last pair when an odd count was asked for).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

namespace strumpack {
namespace HSS {

class LinearNormal {
 public:
  explicit LinearNormal(std::uint32_t seed = 0) { state_ = seed % M ? seed % M : 1u; }   // (linear_congruential_engine::seed)

  // out[0 .. count) = the next `count` values of the stream.  parallel_for(n, fn) runs fn(0..n-1), possibly concurrently.
  void fill(double* out, std::size_t count, const std::function<void(std::size_t, const std::function<void(std::size_t)>&)>& parallel_for) {
    std::size_t done = 0;
    if (count && has_saved_) {
      out[done++] = saved_ * 1.0 + 0.0;
      has_saved_ = false;
    }
    while (done < count) {
      const std::size_t pairs = (count - done + 1) / 2;   // accepted attempts still needed
      // attempts of this sweep: the expected number (acceptance pi / 4) plus a margin; a short sweep just repeats
      const std::size_t attempts = (std::size_t)((double)pairs * 1.2740) + 4096;
      const std::size_t chunk = 16384;
      const std::size_t nchunk = (attempts + chunk - 1) / chunk;
      // accepted pairs of chunk c at raw[2 * c * chunk ...] (one uninitialised block: per-chunk vectors of this size would
      // each be a separate mapping, and unmapping them from many threads serialises on the address space)
      std::unique_ptr<double[]> raw(new double[2 * nchunk * chunk]);
      std::vector<std::size_t> cnt(nchunk, 0);
      const std::uint32_t base = state_;
      parallel_for(nchunk, [&](std::size_t c) {
        const std::size_t j0 = c * chunk, j1 = std::min(attempts, j0 + chunk);
        std::uint32_t s = jump(base, 4 * (std::uint64_t)j0);
        double* v = raw.get() + 2 * j0;
        std::size_t n = 0;
        for (std::size_t j = j0; j < j1; j++) {
          double x, y;
          if (attempt(s, x, y)) { v[n++] = y; v[n++] = x; }   // (the order the library returns them in)
        }
        cnt[c] = n / 2;
      });
      // accepted pairs in attempt order; stop inside the chunk that completes the request
      std::vector<std::size_t> first(nchunk + 1, 0);
      for (std::size_t c = 0; c < nchunk; c++) first[c + 1] = first[c] + cnt[c];
      const std::size_t take = std::min(pairs, first[nchunk]);
      double* dst = out + done;
      const std::size_t room = count - done;   // values still to write (the last pair may give one only)
      parallel_for(nchunk, [&](std::size_t c) {
        if (first[c] >= take) return;
        const std::size_t np = std::min(cnt[c], take - first[c]);
        const double* v = raw.get() + 2 * c * chunk;
        for (std::size_t p = 0; p < np; p++) {
          const std::size_t o = 2 * (first[c] + p);
          dst[o] = v[2 * p] * 1.0 + 0.0;   // (the unit parameters, applied as the library applies them: -0 becomes +0)
          if (o + 1 < room) dst[o + 1] = v[2 * p + 1] * 1.0 + 0.0;
        }
      });
      if (take == pairs) {
        // the engine stops behind the attempt that produced the last pair taken: find it in its chunk
        std::size_t c = 0;
        while (first[c + 1] < take) c++;
        const std::size_t want = take - first[c];   // that many accepted attempts of chunk c
        std::size_t j = c * chunk, got = 0;
        std::uint32_t s = jump(base, 4 * (std::uint64_t)j);
        double x = 0, y = 0;
        while (got < want) { if (attempt(s, x, y)) got++; j++; }
        state_ = s;
        const std::size_t wrote = std::min(room, 2 * take);
        if (wrote < 2 * take) { has_saved_ = true; saved_ = x; }   // odd count: x * mult of the last pair is kept
        done += wrote;
      } else {
        state_ = jump(base, 4 * (std::uint64_t)attempts);
        done += 2 * take;
      }
    }
  }

  std::uint32_t state() const { return state_; }

 private:
  static constexpr std::uint32_t A = 48271u, M = 2147483647u;
  static std::uint32_t jump(std::uint32_t s, std::uint64_t k) {
    std::uint64_t r = s, b = A;
    while (k) {
      if (k & 1) r = r * b % M;
      b = b * b % M;
      k >>= 1;
    }
    return (std::uint32_t)r;
  }
  // generate_canonical<double, 53>(minstd_rand&) of libstdc++ (bits/random.tcc): sum of two draws in base R = max - min + 1
  static double canonical(std::uint32_t& s) {
    const double R1 = 2147483646.0;
    const double R2 = (double)(2147483646.0L * 2147483646.0L);
    s = (std::uint32_t)((std::uint64_t)s * A % M);
    double sum = 0.0;
    sum += (double)(s - 1u) * 1.0;
    s = (std::uint32_t)((std::uint64_t)s * A % M);
    sum += (double)(s - 1u) * R1;
    double ret = sum / R2;
    if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
    return ret;
  }
  // one attempt of normal_distribution::operator() (bits/random.tcc): true = accepted, x and y multiplied by mult
  static bool attempt(std::uint32_t& s, double& x, double& y) {
    x = 2.0 * canonical(s) - 1.0;
    y = 2.0 * canonical(s) - 1.0;
    const double r2 = x * x + y * y;
    if (r2 > 1.0 || r2 == 0.0) return false;
    const double mult = std::sqrt(-2 * std::log(r2) / r2);
    x = x * mult;
    y = y * mult;
    return true;
  }

  std::uint32_t state_ = 1;
  bool has_saved_ = false;
  double saved_ = 0.;   // x * mult of the last pair, not yet scaled by the (unit) parameters
};

}  // namespace HSS
}  // namespace strumpack
