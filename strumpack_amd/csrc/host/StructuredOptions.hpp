// structured::StructuredOptions<T> and Type (reference: structured/StructuredOptions.hpp:56-162,
// StructuredOptions.cpp:53-60).  Same setters/getters, defaults and --structured_* flags.
#pragma once
#include <complex>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

namespace strumpack {
namespace structured {

enum class Type : int { HSS = 0, BLR, HODLR, HODBF, BUTTERFLY, LR, LOSSY, LOSSLESS };
inline std::string get_name(Type a) {
  switch (a) {
    case Type::HSS: return "HSS"; case Type::BLR: return "BLR"; case Type::HODLR: return "HODLR";
    case Type::HODBF: return "HODBF"; case Type::BUTTERFLY: return "BUTTERFLY"; case Type::LR: return "LR";
    case Type::LOSSY: return "LOSSY"; case Type::LOSSLESS: return "LOSSLESS";
  }
  return "UNKNOWN";
}

template <typename real_t> inline real_t default_structured_rel_tol() { return real_t(1e-4); }
template <typename real_t> inline real_t default_structured_abs_tol() { return real_t(1e-10); }
template <> inline float default_structured_rel_tol() { return 1e-2f; }
template <> inline float default_structured_abs_tol() { return 1e-5f; }   // structured/StructuredOptions.hpp:52-54

namespace detail {
// "--name value" / "--name=value" scanner shared by the option classes (the reference uses
// getopt_long_only; unknown flags are ignored there too)
inline bool match_flag(int argc, const char* const* argv, int& i, const char* name, std::string& val, bool has_arg) {
  const char* a = argv[i];
  while (*a == '-') a++;
  std::size_t n = std::strlen(name);
  if (std::strncmp(a, name, n) != 0) return false;
  if (a[n] == '=') { val = a + n + 1; return has_arg; }
  if (a[n] != '\0') return false;
  if (!has_arg) return true;
  if (i + 1 < argc) { val = argv[++i]; return true; }
  return false;
}
}  // namespace detail

// real type of a scalar type (reference: misc/RandomWrapper.hpp RealType)
template <typename T> struct RealType { using value_type = T; };
template <typename R> struct RealType<std::complex<R>> { using value_type = R; };

template <typename scalar_t> class StructuredOptions {
  using real_t = typename RealType<scalar_t>::value_type;

 public:
  StructuredOptions() {}
  StructuredOptions(Type type) : type_(type) {}
  virtual ~StructuredOptions() {}
  void set_rel_tol(real_t rel_tol) { rel_tol_ = rel_tol; }
  void set_abs_tol(real_t abs_tol) { abs_tol_ = abs_tol; }
  void set_leaf_size(int leaf_size) { leaf_size_ = leaf_size; }
  void set_pivot_threshold(real_t thresh) { pivot_ = thresh; }
  void set_max_rank(int max_rank) { max_rank_ = max_rank; }
  void set_type(Type a) { type_ = a; }
  void set_verbose(bool verbose) { verbose_ = verbose; }
  real_t rel_tol() const { return rel_tol_; }
  real_t abs_tol() const { return abs_tol_; }
  real_t pivot_threshold() const { return pivot_; }
  int leaf_size() const { return leaf_size_; }
  int max_rank() const { return max_rank_; }
  Type type() const { return type_; }
  bool verbose() const { return verbose_; }
  virtual void set_from_command_line(int argc, const char* const* argv) {
    for (int i = 1; i < argc; i++) {
      std::string v;
      if (detail::match_flag(argc, argv, i, "structured_rel_tol", v, true)) set_rel_tol(std::atof(v.c_str()));
      else if (detail::match_flag(argc, argv, i, "structured_abs_tol", v, true)) set_abs_tol(std::atof(v.c_str()));
      else if (detail::match_flag(argc, argv, i, "structured_leaf_size", v, true)) set_leaf_size(std::atoi(v.c_str()));
      else if (detail::match_flag(argc, argv, i, "structured_max_rank", v, true)) set_max_rank(std::atoi(v.c_str()));
      else if (detail::match_flag(argc, argv, i, "structured_type", v, true)) {
        for (int t = 0; t <= int(Type::LOSSLESS); t++) if (v == get_name(Type(t))) set_type(Type(t));
      } else if (detail::match_flag(argc, argv, i, "structured_verbose", v, false)) set_verbose(true);
      else if (detail::match_flag(argc, argv, i, "structured_quiet", v, false)) set_verbose(false);
    }
  }
  virtual void describe_options() const {
    std::cout << "# Structured Options:\n#   --structured_rel_tol real_t (default " << rel_tol() << ")\n"
              << "#   --structured_abs_tol real_t (default " << abs_tol() << ")\n"
              << "#   --structured_leaf_size int (default " << leaf_size() << ")\n"
              << "#   --structured_max_rank int (default " << max_rank() << ")\n"
              << "#   --structured_type [HSS]  (this build implements the HSS hot path)\n"
              << "#   --structured_verbose or -v / --structured_quiet or -q" << std::endl;
  }

 protected:
  Type type_ = Type::BLR;
  real_t rel_tol_ = default_structured_rel_tol<real_t>();
  real_t abs_tol_ = default_structured_abs_tol<real_t>();
  real_t pivot_ = -1.;
  int leaf_size_ = 128;
  int max_rank_ = 5000;
  bool verbose_ = true;
};

// the same options for the double-precision engine that carries the float / complex instantiations
template <typename T> StructuredOptions<double> to_double_options(const StructuredOptions<T>& o) {
  StructuredOptions<double> d(o.type());
  d.set_rel_tol(o.rel_tol()); d.set_abs_tol(o.abs_tol()); d.set_leaf_size(o.leaf_size());
  d.set_pivot_threshold(o.pivot_threshold()); d.set_max_rank(o.max_rank()); d.set_verbose(o.verbose());
  return d;
}

}  // namespace structured
}  // namespace strumpack
