// HSS::HSSOptions<T> (reference: HSS/HSSOptions.hpp:153-493, HSSOptions.cpp:100-235): the knobs that
// change the arithmetic of the HSS hot path, with the reference's names, defaults and --hss_* flags.
// One addition: RandomEngine::PHILOX draws the sketching matrix on the device (counter-based
// Philox4x32-10); LINEAR / MERSENNE reproduce the reference's host generators bit-for-bit.
#pragma once
#include <cassert>

#include "Clustering.hpp"
#include "StructuredOptions.hpp"

namespace strumpack {
namespace random {
enum class RandomEngine { LINEAR, MERSENNE, PHILOX };
enum class RandomDistribution { NORMAL, UNIFORM };
inline std::string get_name(RandomEngine e) {
  switch (e) { case RandomEngine::LINEAR: return "minstd_rand"; case RandomEngine::MERSENNE: return "mt19937"; case RandomEngine::PHILOX: return "philox (device)"; }
  return "unknown";
}
inline std::string get_name(RandomDistribution d) { return d == RandomDistribution::NORMAL ? "normal(0,1)" : "uniform[0,1]"; }
}  // namespace random

namespace HSS {

enum class CompressionAlgorithm { ORIGINAL, STABLE, HARD_RESTART };
enum class CompressionSketch { GAUSSIAN, SJLT };
// how the nnz nonzeros of a row of the SJLT sketching matrix are placed (reference HSSOptions.hpp:128-133):
// CHUNK = one in each of nnz equal column chunks, PERM = the first nnz entries of a random permutation
enum class SJLTAlgo { CHUNK, PERM };
inline std::string get_name(CompressionAlgorithm a) {
  switch (a) { case CompressionAlgorithm::ORIGINAL: return "original"; case CompressionAlgorithm::STABLE: return "stable"; case CompressionAlgorithm::HARD_RESTART: return "hard_restart"; }
  return "unknown";
}

// neighbour search of the kernel-matrix compression (extension): EXACT = all-pairs k nearest neighbours on the device
// (default), ANN = the reference's randomized projection-tree search on the host (identical lists, hence an HSS
// matrix identical to the reference's)
enum class NeighborSearch { EXACT, ANN };

template <typename real_t> inline real_t default_HSS_rel_tol() { return real_t(1e-2); }
template <typename real_t> inline real_t default_HSS_abs_tol() { return real_t(1e-8); }

template <typename scalar_t> class HSSOptions : public structured::StructuredOptions<scalar_t> {
  using real_t = scalar_t;

 public:
  HSSOptions() : structured::StructuredOptions<scalar_t>(structured::Type::HSS) { set_defaults(); }
  // conversion keeps the structured values (reference HSSOptions.hpp:166-169)
  HSSOptions(const structured::StructuredOptions<scalar_t>& sopts) : structured::StructuredOptions<scalar_t>(sopts) {
    this->type_ = structured::Type::HSS;
  }
  void set_d0(int d0) { assert(d0 > 0); d0_ = d0; }
  void set_dd(int dd) { assert(dd > 0); dd_ = dd; }
  void set_p(int p) { assert(p >= 0); p_ = p; }
  void set_random_engine(random::RandomEngine e) { random_engine_ = e; }
  void set_random_distribution(random::RandomDistribution d) { random_distribution_ = d; }
  void set_compression_algorithm(CompressionAlgorithm a) { compress_algo_ = a; }
  void set_compression_sketch(CompressionSketch a) { compress_sketch_ = a; }
  // SJLT sketch (reference HSSOptions.hpp:195-204, :247): nonzeros per row of the first d0+dd columns / of each
  // further block of dd columns, and their placement
  void set_nnz0(int nnz0) { assert(nnz0 > 0); nnz0_ = nnz0; }
  void set_nnz(int nnz) { assert(nnz > 0); nnz_ = nnz; }
  void set_SJLT_algo(SJLTAlgo a) { sjlt_algo_ = a; }
  void set_user_defined_random(bool u) { user_defined_random_ = u; }
  // extension: the caller will factor -- the ULV factorization of each tree level is enqueued on a second stream as soon as
  // the compression has settled the level (DeviceHSS: EngineOptions::factor_ahead); factor() then only waits
  // extension: the operand is symmetric by the caller's word (1: trusted, 2: checked on a sample): A^T R = A R, one sketch GEMM
  void set_symmetric_operand(int s) { symmetric_ = s; }
  int symmetric_operand() const { return symmetric_; }
  void set_factor_ahead(bool f) { factor_ahead_ = f; }
  bool factor_ahead() const { return factor_ahead_; }
  void set_synchronized_compression(bool sync) { sync_ = sync; }
  void set_log_ranks(bool log_ranks) { log_ranks_ = log_ranks; }
  // kernel-matrix construction (reference HSSOptions.hpp:258-283)
  void set_clustering_algorithm(ClusteringAlgorithm a) { clustering_algorithm_ = a; }
  void set_approximate_neighbors(int neighbors) { approximate_neighbors_ = neighbors; }
  void set_ann_iterations(int iters) { assert(iters > 0); ann_iterations_ = iters; }
  void set_neighbor_search(NeighborSearch s) { neighbor_search_ = s; }
  int d0() const { return d0_; }
  int dd() const { return dd_; }
  int p() const { return p_; }
  random::RandomEngine random_engine() const { return random_engine_; }
  random::RandomDistribution random_distribution() const { return random_distribution_; }
  CompressionAlgorithm compression_algorithm() const { return compress_algo_; }
  CompressionSketch compression_sketch() const { return compress_sketch_; }
  int nnz0() const { return nnz0_; }
  int nnz() const { return nnz_; }
  SJLTAlgo SJLT_algo() const { return sjlt_algo_; }
  bool user_defined_random() const { return user_defined_random_; }
  bool synchronized_compression() const { return sync_; }
  bool log_ranks() const { return log_ranks_; }
  ClusteringAlgorithm clustering_algorithm() const { return clustering_algorithm_; }
  int approximate_neighbors() const { return approximate_neighbors_; }
  int ann_iterations() const { return ann_iterations_; }
  NeighborSearch neighbor_search() const { return neighbor_search_; }

  void set_from_command_line(int argc, const char* const* argv) override {
    using structured::detail::match_flag;
    for (int i = 1; i < argc; i++) {
      std::string v;
      if (match_flag(argc, argv, i, "hss_rel_tol", v, true)) this->set_rel_tol(std::atof(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_abs_tol", v, true)) this->set_abs_tol(std::atof(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_leaf_size", v, true)) this->set_leaf_size(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_d0", v, true)) set_d0(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_dd", v, true)) set_dd(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_p", v, true)) set_p(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_max_rank", v, true)) this->set_max_rank(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_random_distribution", v, true)) {
        if (v == "normal") set_random_distribution(random::RandomDistribution::NORMAL);
        else if (v == "uniform") set_random_distribution(random::RandomDistribution::UNIFORM);
        else std::cerr << "# WARNING: random number distribution not recognized, use 'normal' or 'uniform'" << std::endl;
      } else if (match_flag(argc, argv, i, "hss_random_engine", v, true)) {
        if (v == "linear") set_random_engine(random::RandomEngine::LINEAR);
        else if (v == "mersenne") set_random_engine(random::RandomEngine::MERSENNE);
        else if (v == "philox") set_random_engine(random::RandomEngine::PHILOX);
        else std::cerr << "# WARNING: random number engine not recognized, use 'linear', 'mersenne' or 'philox'" << std::endl;
      } else if (match_flag(argc, argv, i, "hss_compression_algorithm", v, true)) {
        if (v == "original") set_compression_algorithm(CompressionAlgorithm::ORIGINAL);
        else if (v == "stable") set_compression_algorithm(CompressionAlgorithm::STABLE);
        else if (v == "hard_restart") set_compression_algorithm(CompressionAlgorithm::HARD_RESTART);
        else std::cerr << "# WARNING: compression algorithm not recognized, use 'original', 'stable' or 'hard_restart'" << std::endl;
      } else if (match_flag(argc, argv, i, "hss_compression_sketch", v, true)) {
        if (v == "Gaussian" || v == "gaussian") set_compression_sketch(CompressionSketch::GAUSSIAN);
        else if (v == "SJLT" || v == "sjlt") set_compression_sketch(CompressionSketch::SJLT);
        else std::cerr << "# WARNING: compression sketch not recognized, use 'Gaussian', or 'SJLT'." << std::endl;
      } else if (match_flag(argc, argv, i, "hss_SJLT_algo", v, true)) {
        if (v == "chunk") set_SJLT_algo(SJLTAlgo::CHUNK);
        else if (v == "perm") set_SJLT_algo(SJLTAlgo::PERM);
        else std::cerr << "# WARNING: SJLT algorithm not recognized, use 'chunk' or 'perm'." << std::endl;
      } else if (match_flag(argc, argv, i, "hss_nnz0", v, true)) set_nnz0(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_nnz", v, true)) set_nnz(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_user_defined_random", v, false)) set_user_defined_random(true);
      else if (match_flag(argc, argv, i, "hss_factor_ahead", v, false)) set_factor_ahead(true);
      else if (match_flag(argc, argv, i, "hss_enable_sync", v, false)) set_synchronized_compression(true);
      else if (match_flag(argc, argv, i, "hss_disable_sync", v, false)) set_synchronized_compression(false);
      else if (match_flag(argc, argv, i, "hss_log_ranks", v, false)) set_log_ranks(true);
      else if (match_flag(argc, argv, i, "hss_clustering_algorithm", v, true)) set_clustering_algorithm(get_clustering_algorithm(v));
      else if (match_flag(argc, argv, i, "hss_approximate_neighbors", v, true)) set_approximate_neighbors(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_ann_iterations", v, true)) set_ann_iterations(std::atoi(v.c_str()));
      else if (match_flag(argc, argv, i, "hss_neighbor_search", v, true)) {
        if (v == "exact") set_neighbor_search(NeighborSearch::EXACT);
        else if (v == "ann") set_neighbor_search(NeighborSearch::ANN);
        else std::cerr << "# WARNING: neighbour search not recognized, use 'exact' or 'ann'" << std::endl;
      }
      else if (match_flag(argc, argv, i, "hss_verbose", v, false) || std::string(argv[i]) == "-v") this->set_verbose(true);
      else if (match_flag(argc, argv, i, "hss_quiet", v, false) || std::string(argv[i]) == "-q") this->set_verbose(false);
    }
  }
  void describe_options() const override {
    std::cout << "# HSS Options:\n#   --hss_rel_tol real_t (default " << this->rel_tol() << ")\n#   --hss_abs_tol real_t (default " << this->abs_tol()
              << ")\n#   --hss_leaf_size int (default " << this->leaf_size() << ")\n#   --hss_d0 int (default " << d0() << ")\n#   --hss_dd int (default " << dd()
              << ")\n#   --hss_p int (default " << p() << ")\n#   --hss_max_rank int (default " << this->max_rank()
              << ")\n#   --hss_random_distribution normal|uniform\n#   --hss_random_engine linear|mersenne|philox\n"
              << "#   --hss_compression_algorithm original|stable|hard_restart\n#   --hss_compression_sketch Gaussian|SJLT\n"
              << "#   --hss_SJLT_algo chunk|perm\n#   --hss_nnz0 int (default " << nnz0() << ")\n#   --hss_nnz int (default " << nnz() << ")\n"
              << "#   --hss_clustering_algorithm natural|2means|kdtree|pca|cobble (default " << get_name(clustering_algorithm()) << ")\n#   --hss_approximate_neighbors int (default " << approximate_neighbors()
              << ")\n#   --hss_ann_iterations int (default " << ann_iterations() << ")\n#   --hss_neighbor_search exact|ann (default exact: all pairs on the device)\n"
              << "#   --hss_user_defined_random  --hss_enable_sync  --hss_disable_sync  --hss_log_ranks\n#   --hss_verbose or -v   --hss_quiet or -q" << std::endl;
  }

 private:
  void set_defaults() {  // reference HSSOptions.hpp:484-490
    this->rel_tol_ = default_HSS_rel_tol<real_t>();
    this->abs_tol_ = default_HSS_abs_tol<real_t>();
    this->leaf_size_ = 512;
    this->max_rank_ = 50000;
  }
  int d0_ = 128, dd_ = 64, p_ = 10;
  random::RandomEngine random_engine_ = random::RandomEngine::LINEAR;
  random::RandomDistribution random_distribution_ = random::RandomDistribution::NORMAL;
  CompressionAlgorithm compress_algo_ = CompressionAlgorithm::STABLE;
  CompressionSketch compress_sketch_ = CompressionSketch::GAUSSIAN;
  int nnz0_ = 4, nnz_ = 4;
  SJLTAlgo sjlt_algo_ = SJLTAlgo::CHUNK;
  bool user_defined_random_ = false, sync_ = false, log_ranks_ = false, factor_ahead_ = false;
  int symmetric_ = 0;
  ClusteringAlgorithm clustering_algorithm_ = ClusteringAlgorithm::TWO_MEANS;
  int approximate_neighbors_ = 64, ann_iterations_ = 5;
  NeighborSearch neighbor_search_ = NeighborSearch::EXACT;
};

}  // namespace HSS
}  // namespace strumpack
