// flatten.cpp — World -> SoA buffers for the device (RtcSceneDesc).
//
// `Shape::set_transform` already pushed every group transform down to the leaves (shape.rs:207-218), so the
// world is a depth-first list of leaves, each carrying its cached `transform_inverse` (shape.rs:45) — the
// matrix the reference recomputes per ray per leaf (shape.rs:249-253) is a flatten-time constant here.
// Groups survive only as (parent, Bounds::new box) so that the device can apply the same bounding-box gate
// the reference's Group arm applies (shape.rs:399-425).  Identical transforms / materials / patterns are
// stored once (an OBJ mesh has exactly one of each).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "rtc_host.hpp"

namespace rtc::host {

RtcSceneDesc FlatScene::desc() const {
  RtcSceneDesc d{};
  std::memcpy(d.light_position, light_position, sizeof d.light_position);
  std::memcpy(d.light_intensity, light_intensity, sizeof d.light_intensity);
  d.n_transforms = (uint32_t)(transform_inverse.size() / 16);
  d.transform_inverse = transform_inverse.data();
  d.n_materials = (uint32_t)materials.size();
  d.materials = materials.data();
  d.n_patterns = (uint32_t)patterns.size();
  d.patterns = patterns.data();
  d.n_groups = (uint32_t)group_parent.size();
  d.group_parent = group_parent.data();
  d.group_bounds_min = group_bounds_min.data();
  d.group_bounds_max = group_bounds_max.data();
  d.n_leaves = (uint32_t)leaf_kind.size();
  d.leaf_kind = leaf_kind.data();
  d.leaf_transform = leaf_transform.data();
  d.leaf_material = leaf_material.data();
  d.leaf_group = leaf_group.data();
  d.leaf_class = leaf_class.empty() ? nullptr : leaf_class.data();  // NULL: every leaf is its own class
  d.leaf_params = leaf_params.data();
  d.leaf_triangle = leaf_triangle.data();
  d.n_triangles = (uint32_t)(tri_p1.size() / 3);
  d.tri_p1 = tri_p1.data();
  d.tri_e1 = tri_e1.data();
  d.tri_e2 = tri_e2.data();
  d.tri_normal = tri_normal.data();
  return d;
}

namespace {

template <size_t N>
std::array<uint64_t, N> bits(const double* p) {
  std::array<uint64_t, N> k;
  std::memcpy(k.data(), p, N * sizeof(double));
  return k;
}

struct Flattener {
  FlatScene out;
  std::vector<const Shape*> leaves;
  std::map<std::array<uint64_t, 16>, uint32_t> xf_ids;
  std::map<std::array<uint64_t, 34>, uint32_t> mat_ids;  // material fields + pattern payload, bit-exact

  // consecutive leaves nearly always share transform and material (a mesh has one of each): remember the last key
  std::array<uint64_t, 16> last_xf_key{};
  std::array<uint64_t, 34> last_mat_key{};
  uint32_t last_xf = 0, last_mat = 0;
  bool have_last_xf = false, have_last_mat = false;

  uint32_t transform_id(const Matrix4& inv) {
    auto key = bits<16>(inv.m.data());
    if (have_last_xf && key == last_xf_key) return last_xf;
    last_xf_key = key, have_last_xf = true;
    return last_xf = transform_id_slow(inv, key);
  }
  uint32_t transform_id_slow(const Matrix4& inv, const std::array<uint64_t, 16>& key) {
    auto it = xf_ids.find(key);
    if (it != xf_ids.end()) return it->second;
    uint32_t id = (uint32_t)(out.transform_inverse.size() / 16);
    out.transform_inverse.insert(out.transform_inverse.end(), inv.m.begin(), inv.m.end());
    xf_ids.emplace(key, id);
    return id;
  }

  uint32_t material_id(const Material& m) {
    double flat[34] = {m.color.red, m.color.green, m.color.blue, m.ambient,      m.diffuse,
                       m.specular,  m.shininess,   m.reflective, m.transparency, m.refractive_index};
    if (m.pattern) {
      const Pattern& p = *m.pattern;
      flat[10] = (double)p.kind();
      flat[11] = p.a().red, flat[12] = p.a().green, flat[13] = p.a().blue;
      flat[14] = p.b().red, flat[15] = p.b().green, flat[16] = p.b().blue;
      std::memcpy(&flat[17], p.transform_inverse().m.data(), 16 * sizeof(double));
      flat[33] = 1.0;
    }
    auto key = bits<34>(flat);
    if (have_last_mat && key == last_mat_key) return last_mat;
    last_mat_key = key, have_last_mat = true;
    return last_mat = material_id_slow(m, key);
  }
  uint32_t material_id_slow(const Material& m, const std::array<uint64_t, 34>& key) {
    auto it = mat_ids.find(key);
    if (it != mat_ids.end()) return it->second;
    RtcMaterial r{};
    r.color[0] = m.color.red, r.color[1] = m.color.green, r.color[2] = m.color.blue;
    r.ambient = m.ambient, r.diffuse = m.diffuse, r.specular = m.specular, r.shininess = m.shininess;
    r.reflective = m.reflective, r.transparency = m.transparency, r.refractive_index = m.refractive_index;
    r.pattern = -1;
    if (m.pattern) {
      const Pattern& p = *m.pattern;
      RtcPattern rp{};
      rp.kind = p.kind();
      rp.a[0] = p.a().red, rp.a[1] = p.a().green, rp.a[2] = p.a().blue;
      rp.b[0] = p.b().red, rp.b[1] = p.b().green, rp.b[2] = p.b().blue;
      std::memcpy(rp.transform_inverse, p.transform_inverse().m.data(), sizeof rp.transform_inverse);
      r.pattern = (int32_t)out.patterns.size();
      out.patterns.push_back(rp);
    }
    uint32_t id = (uint32_t)out.materials.size();
    out.materials.push_back(r);
    mat_ids.emplace(key, id);
    return id;
  }

  // Returns Bounds::new(s) (bounds.rs:11-140).  A group's box is assembled from the boxes its children return, with
  // the arithmetic of Shape::bounds() (eight corners of each child box through the child's transform, bounds.rs:62-120):
  // the same values as Bounds::new(group), without walking every subtree once per enclosing group.
  Bounds visit(const Shape& s, int32_t parent_group) {
    if (s.kind() == ShapeKind::Group) {
      const size_t gid = out.group_parent.size();
      out.group_parent.push_back(parent_group);
      out.group_bounds_min.insert(out.group_bounds_min.end(), 3, 0.0);
      out.group_bounds_max.insert(out.group_bounds_max.end(), 3, 0.0);
      Bounds b{Tuple::point(0, 0, 0), Tuple::point(0, 0, 0)};  // seeded with the origin (reference quirk)
      for (const Shape& c : *s.shapes()) {
        const Bounds pb = visit(c, (int32_t)gid);
        const Matrix4& t = c.transform();
        const double xs[2] = {pb.min.x, pb.max.x}, ys[2] = {pb.min.y, pb.max.y}, zs[2] = {pb.min.z, pb.max.z};
        for (int ix = 0; ix < 2; ++ix)
          for (int iy = 0; iy < 2; ++iy)
            for (int iz = 0; iz < 2; ++iz) {
              const Tuple p = t * Tuple::point(xs[ix], ys[iy], zs[iz]);
              b.min.x = std::fmin(b.min.x, p.x), b.min.y = std::fmin(b.min.y, p.y), b.min.z = std::fmin(b.min.z, p.z);
              b.max.x = std::fmax(b.max.x, p.x), b.max.y = std::fmax(b.max.y, p.y), b.max.z = std::fmax(b.max.z, p.z);
            }
      }
      out.group_bounds_min[3 * gid] = b.min.x, out.group_bounds_min[3 * gid + 1] = b.min.y, out.group_bounds_min[3 * gid + 2] = b.min.z;
      out.group_bounds_max[3 * gid] = b.max.x, out.group_bounds_max[3 * gid + 1] = b.max.y, out.group_bounds_max[3 * gid + 2] = b.max.z;
      return b;
    }
    uint8_t kind = 0;
    switch (s.kind()) {
      case ShapeKind::Sphere: kind = RTC_SPHERE; break;
      case ShapeKind::Plane: kind = RTC_PLANE; break;
      case ShapeKind::Cube: kind = RTC_CUBE; break;
      case ShapeKind::Cylinder: kind = RTC_CYLINDER; break;
      case ShapeKind::Cone: kind = RTC_CONE; break;
      case ShapeKind::Triangle: kind = RTC_TRIANGLE; break;
      default: break;
    }
    leaves.push_back(&s);
    out.leaf_kind.push_back(kind);
    out.leaf_transform.push_back(transform_id(s.transform_inverse()));
    out.leaf_material.push_back(material_id(s.get_material()));
    out.leaf_group.push_back(parent_group);
    out.leaf_params.insert(out.leaf_params.end(), {s.minimum, s.maximum, s.capped ? 1.0 : 0.0});
    if (kind == RTC_TRIANGLE) {
      out.leaf_triangle.push_back((uint32_t)(out.tri_p1.size() / 3));
      out.tri_p1.insert(out.tri_p1.end(), {s.p1.x, s.p1.y, s.p1.z});
      out.tri_e1.insert(out.tri_e1.end(), {s.e1.x, s.e1.y, s.e1.z});
      out.tri_e2.insert(out.tri_e2.end(), {s.e2.x, s.e2.y, s.e2.z});
      out.tri_normal.insert(out.tri_normal.end(), {s.normal.x, s.normal.y, s.normal.z});
    } else {
      out.leaf_triangle.push_back(0);
    }
    return s.bounds();
  }

  // leaf_class[i] = first leaf j <= i with *leaf_j == *leaf_i (Shape's value equality, shape.rs:638-646).
  void classes() {
    const size_t n = leaves.size();
    out.leaf_class.resize(n);
    std::vector<std::pair<double, uint32_t>> tri_by_x;  // candidates must agree on p1.x within EPSILON
    std::vector<uint32_t> prims;
    for (size_t i = 0; i < n; ++i) {
      if (leaves[i]->kind() == ShapeKind::Triangle)
        tri_by_x.emplace_back(leaves[i]->p1.x, (uint32_t)i);
      else
        prims.push_back((uint32_t)i);
    }
    std::sort(tri_by_x.begin(), tri_by_x.end());
    for (size_t i = 0; i < n; ++i) {
      uint32_t best = (uint32_t)i;
      const Shape& s = *leaves[i];
      if (s.kind() == ShapeKind::Triangle) {
        auto lo = std::lower_bound(tri_by_x.begin(), tri_by_x.end(), std::make_pair(s.p1.x - EPSILON, 0u));
        for (auto it = lo; it != tri_by_x.end() && it->first < s.p1.x + EPSILON; ++it)
          if (it->second < best && *leaves[it->second] == s) best = it->second;
      } else {
        for (uint32_t j : prims) {
          if (j >= best) break;
          if (*leaves[j] == s) {
            best = j;
            break;
          }
        }
      }
      out.leaf_class[i] = best;
    }
  }
};

}  // namespace

FlatScene World::flatten(bool always_classes) const {
  Flattener f;
  size_t n_leaves = 0;
  for (const Shape& s : objects) n_leaves += s.leaf_count();
  f.out.leaf_kind.reserve(n_leaves), f.out.leaf_transform.reserve(n_leaves), f.out.leaf_material.reserve(n_leaves);
  f.out.leaf_group.reserve(n_leaves), f.out.leaf_triangle.reserve(n_leaves), f.out.leaf_params.reserve(3 * n_leaves);
  f.out.tri_p1.reserve(3 * n_leaves), f.out.tri_e1.reserve(3 * n_leaves), f.out.tri_e2.reserve(3 * n_leaves);
  f.out.tri_normal.reserve(3 * n_leaves), f.leaves.reserve(n_leaves);
  f.out.light_position[0] = light.position.x, f.out.light_position[1] = light.position.y, f.out.light_position[2] = light.position.z;
  f.out.light_intensity[0] = light.intensity.red, f.out.light_intensity[1] = light.intensity.green, f.out.light_intensity[2] = light.intensity.blue;
  for (const Shape& s : objects) f.visit(s, -1);
  // value-equality classes are only consumed by prepare_computations' container walk, i.e. when some material is
  // transparent (world.rs:136); without one the table is left out (RtcSceneDesc.leaf_class = NULL)
  bool transparent = always_classes;
  for (const RtcMaterial& m : f.out.materials) transparent = transparent || m.transparency != 0.0;
  if (transparent) f.classes();
  return std::move(f.out);
}

}  // namespace rtc::host
