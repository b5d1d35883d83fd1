// Kernel ridge regression driver in the shape of the reference's examples/dense/KernelRegression.cpp (same command
// line: file d h lambda degree kernel mode, followed by --hss_* options; same classes and calls), written against this
// repository's headers.  Prints the prediction score; exits non-zero if the score is below the bound given in the
// environment (KRR_MIN_SCORE, percent) -- the reference's example has no pass criterion of its own.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "HSSOptions.hpp"
#include "Kernel.hpp"

using namespace strumpack;

static std::vector<double> read_csv(const std::string& name) {
  std::vector<double> v;
  std::ifstream f(name);
  if (!f) { std::cerr << "cannot open " << name << std::endl; std::exit(2); }
  std::string line, cell;
  while (std::getline(f, line)) {
    std::istringstream row(line);
    while (std::getline(row, cell, ',')) v.push_back(std::stod(cell));
  }
  return v;
}

int main(int argc, char* argv[]) {
  std::string filename("./data/susy_10Kn"), mode("test");
  std::size_t d = 8;
  double h = 1.3, lambda = 3.11;
  int p = 1;
  kernel::KernelType ktype = kernel::KernelType::GAUSS;
  std::cout << "# usage: ./KernelRegression file d h lambda degree kernel(Gauss, Laplace) mode(valid, test)" << std::endl;
  if (argc > 1) filename = argv[1];
  if (argc > 2) d = std::stoi(argv[2]);
  if (argc > 3) h = std::stof(argv[3]);
  if (argc > 4) lambda = std::stof(argv[4]);
  if (argc > 5) p = std::stoi(argv[5]);
  if (argc > 6) ktype = kernel::kernel_type(argv[6]);
  if (argc > 7) mode = argv[7];
  std::cout << "# file            = " << filename << "\n# data dimension  = " << d << "\n# kernel h        = " << h
            << "\n# lambda          = " << lambda << "\n# p               = " << p << "\n# kernel type     = " << kernel::get_name(ktype)
            << "\n# validation/test = " << mode << std::endl;

  HSS::HSSOptions<double> hss_opts;
  hss_opts.set_verbose(true);
  hss_opts.set_from_command_line(argc, argv);

  auto training = read_csv(filename + "_train.csv");
  auto testing = read_csv(filename + "_" + mode + ".csv");
  auto train_labels = read_csv(filename + "_train_label.csv");
  auto test_labels = read_csv(filename + "_" + mode + "_label.csv");
  std::size_t n = training.size() / d, m = testing.size() / d;
  if (const char* e = std::getenv("KRR_MAX_POINTS")) {   // test tier: a prefix of the training set
    n = std::min<std::size_t>(n, std::atoi(e));
    training.resize(n * d);
    train_labels.resize(n);
  }
  std::cout << "# training dataset = " << n << " x " << d << "\n# testing dataset  = " << m << " x " << d << std::endl;

  DenseMatrixWrapper<double> training_points(d, n, training.data(), d), test_points(d, m, testing.data(), d);
  auto K = kernel::create_kernel<double>(ktype, training_points, h, lambda, p);
  auto weights = K->fit_HSS(train_labels, hss_opts);
  auto prediction = K->predict(test_points, weights);

  std::size_t incorrect = 0;
  for (std::size_t i = 0; i < m; i++)
    if ((prediction[i] >= 0 && test_labels[i] < 0) || (prediction[i] < 0 && test_labels[i] >= 0)) incorrect++;
  const double score = double(m - incorrect) / m * 100.;
  std::cout << "# prediction score: " << score << "%" << std::endl << "# c-err: " << double(incorrect) / m * 100. << "%" << std::endl;
  if (const char* e = std::getenv("KRR_MIN_SCORE"))
    if (score < std::atof(e)) { std::cout << "# score below " << e << "%" << std::endl; return 1; }
  return 0;
}
