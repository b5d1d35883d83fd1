/* C interface to kernel ridge regression through an HSS approximation of the kernel matrix, computed on the
 * MI355X.  Same entry points as the reference's src/kernel/Kernel.h:43-80 (double precision; the reference's
 * Python wrapper src/python/STRUMPACKKernel.py.in binds exactly these through ctypes).
 *
 *   STRUMPACK_create_kernel_double   Kernel.h:43   train: d x n, one point per column (copied);
 *                                                  type 0 Gauss, 1 Laplace, 2 ANOVA (degree p)
 *   STRUMPACK_kernel_fit_HSS_double  Kernel.h:51   argv carries --hss_* options (default clustering: cobble,
 *                                                  Kernel.cpp:81-83); labels: n values
 *   STRUMPACK_kernel_predict_double  Kernel.h:77   test: d x m; prediction: m values out
 *   STRUMPACK_destroy_kernel_double  Kernel.h:48
 */
#ifndef STRUMPACK_C_KERNEL_HPP
#define STRUMPACK_C_KERNEL_HPP

typedef void* STRUMPACKKernel;

#ifdef __cplusplus
extern "C" {
#endif

STRUMPACKKernel STRUMPACK_create_kernel_double(int n, int d, double* train, double h, double lambda, int p, int type);
void STRUMPACK_destroy_kernel_double(STRUMPACKKernel K);
void STRUMPACK_kernel_fit_HSS_double(STRUMPACKKernel K, double* labels, int argc, char* argv[]);
void STRUMPACK_kernel_predict_double(STRUMPACKKernel K, int m, double* test, double* prediction);

/* ---- extensions (SPX_): introspection for tests / benchmarks --------------------------------------------- */
/* after fit: out[0] compressed (0/1), [1] levels, [2] max rank, [3] memory bytes, [4] neighbour count used,
 * [5] compress us, [6] factor us, [7] solve us */
int SPX_kernel_fit_info(STRUMPACKKernel K, long long* out);
/* pre-order node table of the last fit, 6 ints per node (row_offset, rows, U_rows, U_rank, V_rank, is_leaf);
 * returns the node count */
int SPX_kernel_node_info(STRUMPACKKernel K, int* out, int cap);
/* tests: neighbour lists (k x n ints, 0-based ids in CLUSTER order, column i = point i) used instead of the device
 * search in the first compression round of the next fit */
int SPX_kernel_set_neighbors(STRUMPACKKernel K, int k, const int* ann);
/* the 1-based permutation of the training points chosen by the clustering (n ints) and the weights (n doubles,
 * in permuted order) */
int SPX_kernel_permutation(STRUMPACKKernel K, int* perm);
int SPX_kernel_weights(STRUMPACKKernel K, double* w);
/* binary_tree_clustering on its own (clustering/Clustering.hpp:143-168): algo 0 natural, 1 2means, 2 kdtree,
 * 3 pca, 4 cobble; data (d x n) is reordered in place, perm is 1-based; returns the number of leaves and writes
 * at most cap leaf sizes */
int SPX_clustering(int n, int d, double* data, int algo, int leaf_size, int* perm, int* leaf_sizes, int cap);
/* the same with the median-split partitioners (2 kdtree, 4 cobble) on the device, one launch per tree level
 * (hssk_cluster_median); *status: 0 = done; > 0 = ties at a median / at the farthest point or a long displacement chain were
 * met, -1 = algorithm or dimension not taken by the device form -- data and perm are untouched then and the return value is 0
 * (SPX_clustering, the host form, decides such point sets).  This is what the kernel-matrix constructors call for point sets
 * of STRUMPACK_AMD_CLUSTER_DEVICE_MIN (default 8192) points and more, falling back to the host form on a non-zero status. */
int SPX_clustering_device(int n, int d, double* data, int algo, int leaf_size, int* perm, int* leaf_sizes, int cap, int* status);
/* find_approximate_neighbors on its own (clustering/NeighborSearch.cpp:324-345, host): ann / scores are k x n, column i =
 * the neighbours of point i, nearest first, the point itself included; scores (squared distances) may be NULL */
int SPX_approximate_neighbors(int n, int d, const double* data, int iterations, int k, int* ann, double* scores);

#ifdef __cplusplus
}
#endif
#endif
