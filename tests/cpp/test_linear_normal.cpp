// LinearNormal (the engine's parallel form of the reference's default random stream) against the stream itself:
// std::minstd_rand seeded with 0 under std::normal_distribution<double> (misc/RandomWrapper.hpp:128-191), bit for bit,
// over calls of odd, even and zero length (the saved second value of a pair carries over between calls), and the
// engine state behind the last call.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "LinearNormal.hpp"

using namespace strumpack::HSS;

int main() {
  auto pf = [](std::size_t n, const std::function<void(std::size_t)>& fn) {
    std::atomic<std::size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 6; t++) th.emplace_back([&] { for (std::size_t i = next++; i < n; i = next++) fn(i); });
    for (auto& t : th) t.join();
  };
  std::minstd_rand e(0);
  std::normal_distribution<double> nd;
  LinearNormal g(0);
  const std::size_t counts[] = {1, 2, 3, 0, 7, 100001, 1, 2499999, 2, 1500000, 5, 16384 * 2, 1};
  std::size_t bad = 0, tot = 0;
  for (std::size_t cnt : counts) {
    std::vector<double> a(cnt), b(cnt);
    for (std::size_t i = 0; i < cnt; i++) a[i] = nd(e);
    g.fill(b.data(), cnt, pf);
    for (std::size_t i = 0; i < cnt; i++) if (std::memcmp(&a[i], &b[i], sizeof(double))) bad++;
    tot += cnt;
  }
  // the engines continue alike: same state unless a saved value is pending on both sides (then the next values agree too)
  std::vector<double> a(3), b(3);
  for (auto& v : a) v = nd(e);
  g.fill(b.data(), 3, pf);
  for (int i = 0; i < 3; i++) if (std::memcmp(&a[i], &b[i], sizeof(double))) bad++;
  std::minstd_rand e2(g.state());
  if (!(e == e2)) { std::printf("engine state differs\n"); bad++; }
  std::printf("%s: %zu of %zu values differ\n", bad ? "FAIL" : "PASS", bad, tot + 3);
  return bad != 0;
}
