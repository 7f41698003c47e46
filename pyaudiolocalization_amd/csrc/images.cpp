// images.cpp - image-source generator (host C++; tiny, data-dependent breadth-first search).
//
// Replaces generate_image_sources_iterative (utils.py:67-106) with reflect_point_across_plane
// (utils.py:29-42), distance (utils.py:44-48) and calculate_attenuation (utils.py:50-65):
// reflection orders are expanded breadth first, images are de-duplicated on coordinates rounded
// to `round_decimals`, an image survives when mean(att over mics) > thr and min(att) > thr/2
// (utils.py:99); a pruned image is neither remembered nor expanded.  Discovery order is
// the output order.
#include <cmath>
#include <vector>

#include "../../include/pal_hip.h"

namespace {

struct P3 { double x, y, z; };

// np.add.reduce on a contiguous double vector: first element, then NumPy's pairwise sum of the rest
double pairwise(const double* a, size_t n) {
  if (n < 8) {
    double r = 0.0;
    for (size_t i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    size_t i = 8;
    for (; i + 8 <= n; i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  size_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise(a, n2) + pairwise(a + n2, n - n2);
}

double mean_of(const std::vector<double>& v) {
  if (v.empty()) return NAN;
  return (v[0] + pairwise(v.data() + 1, v.size() - 1)) / double(v.size());
}

double round_dec(double v, double scale) { return std::nearbyint(v * scale) / scale; }   // np.round: rint(x*10^d)/10^d

}  // namespace

extern "C" int pal_image_sources(const double* source, const double* planes, const int32_t* material_id, int K,
                                 const double* absorption, const double* freq_coeff, int n_materials, int max_order,
                                 double frequency, const double* mics, int M, double threshold, int round_decimals,
                                 double* images, int32_t* image_material, int cap, int* count) {
  if (!source || !mics || !count || K < 0 || M < 1 || (K > 0 && (!planes || !material_id)) || (cap > 0 && (!images || !image_material)))
    return PAL_ERR_INVALID;
  for (int k = 0; k < K; ++k) {
    if (material_id[k] >= n_materials) return PAL_ERR_INVALID;
    const double* pl = planes + 4 * k;
    if (pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2] == 0) return PAL_ERR_INVALID;   // utils.py:36-37
  }
  const double scale = std::pow(10.0, round_decimals);
  std::vector<P3> seen, frontier, next;
  auto key = [&](P3 p) { return P3{round_dec(p.x, scale), round_dec(p.y, scale), round_dec(p.z, scale)}; };
  auto known = [&](P3 k) {
    for (const P3& s : seen)
      if (s.x == k.x && s.y == k.y && s.z == k.z) return true;
    return false;
  };
  const P3 src{source[0], source[1], source[2]};
  seen.push_back(key(src));
  frontier.push_back(src);
  int found = 0;
  std::vector<double> att(size_t(M), 0.0);
  for (int order = 1; order <= max_order && !frontier.empty(); ++order) {
    next.clear();
    for (const P3& s : frontier) {
      for (int k = 0; k < K; ++k) {
        const double a = planes[4 * k], b = planes[4 * k + 1], c = planes[4 * k + 2], d = planes[4 * k + 3];
        const double f = 2 * (a * s.x + b * s.y + c * s.z + d) / (a * a + b * b + c * c);
        const P3 img{s.x - a * f, s.y - b * f, s.z - c * f};
        const P3 kk = key(img);
        if (known(kk)) continue;
        const int mat = material_id[k];
        if (mat < 0) {                                        // utils.py:93-96: raised only when reached
          *count = k;
          return PAL_ERR_MATERIAL;
        }
        double lo = INFINITY;
        for (int m = 0; m < M; ++m) {
          const double dx = img.x - mics[3 * m], dy = img.y - mics[3 * m + 1], dz = img.z - mics[3 * m + 2];
          double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
          if (dist < 0.1) dist = 0.1;                                           // utils.py:54-55
          const double v = (1 / dist) * std::exp(-freq_coeff[mat] * frequency * dist) * std::exp(-absorption[mat] * dist);
          att[size_t(m)] = v;
          if (v < lo) lo = v;
        }
        if (mean_of(att) > threshold && lo > threshold / 2) {
          seen.push_back(kk);
          if (found < cap) {
            images[3 * found] = img.x; images[3 * found + 1] = img.y; images[3 * found + 2] = img.z;
            image_material[found] = mat;
          }
          ++found;
          next.push_back(img);
        }
      }
    }
    frontier.swap(next);
  }
  *count = found;
  return found > cap ? PAL_ERR_UNSUPPORTED : PAL_OK;
}
