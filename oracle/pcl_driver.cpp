// pcl_driver.cpp -- TEST INFRASTRUCTURE, and the one file of this repository that cannot be built here: it needs the real PCL 1.7 (the reference
// links the author's fork qianyizh/StanfordPCL, BuildCorrespondence.vcxproj:109,128,137) and FLANN, which neither /root/reference nor this image
// holds.  It exists so that somebody WITH them can close the pin path B lacks (SURVEY.md 8c "parity unpinned", VERDICT round 5 missing 4): it runs
// the calls of CCorresApp::Registration exactly as BuildCorrespondence/CorresApp.cpp:236-312 configures them -- pcl::KdTreeFLANN::nearestKSearch
// (K = 1), pcl::transformPointCloudWithNormals with a Matrix4d, pcl::IterativeClosestPoint + TransformationEstimationPointToPlaneLLS,
// setMaxCorrespondenceDistance / setMaximumIterations( 20 ) / setTransformationEpsilon( 1e-6 ), align( out, guess.cast<float>() ) -- on the
// committed synthetic cases tests/golden/make_golden_pcl.py writes, and prints one JSON object per case: what the two restatements of this
// repository (oracle/icp_oracle.cpp, oracle/stub_corres/er_corres_stub.h) assume about PCL becomes a comparison (tests/test_icp_oracle.py::
// test_restatements_equal_real_pcl_when_its_golden_file_is_present).
//
//   g++ -O2 -std=c++11 oracle/pcl_driver.cpp -o pcl_driver $(pkg-config --cflags --libs pcl_registration-1.7 pcl_kdtree-1.7 pcl_io-1.7 pcl_common-1.7)
//   python tests/golden/make_golden_pcl.py --driver ./pcl_driver          # writes tests/golden/pcl_golden.json
//
// Input: <dir>/cases.txt, one case per line:  name  source.pcd  target.pcd  reg_dist  16 doubles (row-major guess, CorresApp's transformation_)
// Output (stdout): {"pcl_version": "...", "cases": [{...}, ...]} with, per case:
//   nn_index_sha / nn_first: FLANN's nearest neighbour of EVERY transformed source point at the guess (the tie rule shows here),
//   precheck_count: CorresApp.cpp:257-264 (strict <, float squared distance against the double reg_dist_^2),
//   T_hex: icp.getFinalTransformation() as 16 float32 bit patterns, converged, iterations (nr_iterations_ through a subclass), fitness.
#include <pcl/common/transforms.h>
#include <pcl/io/pcd_io.h>
#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/pcl_config.h>
#include <pcl/point_types.h>
#include <pcl/registration/icp.h>
#include <pcl/registration/transformation_estimation_point_to_plane_lls.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

typedef pcl::PointXYZRGBNormal PointT;            // CorresApp.h: the fragments' point type

struct Icp : pcl::IterativeClosestPoint<PointT, PointT> {
  int iterations() const { return nr_iterations_; }
};

// FNV-1a over the index list: no crypto library needed; the Python side computes the same
static uint64_t fnv1a(const std::vector<int>& v) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < v.size(); i++) {
    uint32_t x = (uint32_t)v[i];
    for (int b = 0; b < 4; b++) {
      h ^= (x >> (8 * b)) & 0xffu;
      h *= 1099511628211ull;
    }
  }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: pcl_driver <case directory>\n");
    return 2;
  }
  const std::string dir = argv[1];
  std::ifstream cases((dir + "/cases.txt").c_str());
  if (!cases) {
    fprintf(stderr, "pcl_driver: cannot read %s/cases.txt\n", dir.c_str());
    return 2;
  }
  printf("{\"pcl_version\": \"%s\", \"cases\": [", PCL_VERSION_PRETTY);
  std::string line;
  bool first = true;
  while (std::getline(cases, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream in(line);
    std::string name, fsrc, ftgt;
    double reg_dist;
    Eigen::Matrix4d guess;
    in >> name >> fsrc >> ftgt >> reg_dist;
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) in >> guess(r, c);
    pcl::PointCloud<PointT>::Ptr pcd1(new pcl::PointCloud<PointT>), pcd0(new pcl::PointCloud<PointT>);
    if (pcl::io::loadPCDFile(dir + "/" + fsrc, *pcd1) < 0 || pcl::io::loadPCDFile(dir + "/" + ftgt, *pcd0) < 0) {
      fprintf(stderr, "pcl_driver: cannot load the clouds of case %s\n", name.c_str());
      return 2;
    }
    // ---- CorresApp.cpp:236-264: transform, kd-tree, the pre-check's nearest neighbours ----
    pcl::KdTreeFLANN<PointT> tree;
    const int K = 1;
    std::vector<int> idx(K);
    std::vector<float> sqd(K);
    pcl::PointCloud<PointT>::Ptr transformed(new pcl::PointCloud<PointT>);
    pcl::transformPointCloudWithNormals(*pcd1, *transformed, guess);
    tree.setInputCloud(pcd0);
    std::vector<int> nn(transformed->size(), -1);
    int cnt = 0;
    for (int k = 0; k < (int)transformed->size(); k++) {
      if (tree.nearestKSearch(transformed->points[k], K, idx, sqd) > 0) {
        nn[k] = idx[0];
        if (sqd[0] < reg_dist * reg_dist) cnt++;
      }
    }
    // ---- CorresApp.cpp:295-312 ----
    Icp icp;
    typedef pcl::registration::TransformationEstimationPointToPlaneLLS<PointT, PointT> PointToPlane;
    boost::shared_ptr<PointToPlane> point_to_plane(new PointToPlane);
    icp.setInputCloud(pcd1);
    icp.setInputTarget(pcd0);
    icp.setMaxCorrespondenceDistance(reg_dist);
    icp.setMaximumIterations(20);
    icp.setTransformationEpsilon(1e-6);
    icp.setTransformationEstimation(point_to_plane);
    icp.align(*transformed, guess.cast<float>());
    const Eigen::Matrix4f T = icp.getFinalTransformation();
    printf("%s\n {\"name\": \"%s\", \"source_points\": %d, \"target_points\": %d, \"precheck_count\": %d, \"nn_index_fnv1a\": \"%016llx\", \"nn_first\": [",
           first ? "" : ",", name.c_str(), (int)pcd1->size(), (int)pcd0->size(), cnt, (unsigned long long)fnv1a(nn));
    for (int k = 0; k < 64 && k < (int)nn.size(); k++) printf("%s%d", k ? ", " : "", nn[k]);
    printf("], \"converged\": %s, \"iterations\": %d, \"fitness\": %.17g, \"T_hex\": [", icp.hasConverged() ? "true" : "false", icp.iterations(),
           icp.getFitnessScore());
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        float f = T(r, c);
        uint32_t u;
        memcpy(&u, &f, 4);
        printf("%s\"%08x\"", (r || c) ? ", " : "", u);
      }
    printf("]}");
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
