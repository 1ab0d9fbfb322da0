// example_main.cpp — a C++ program written against rtc_host.hpp the way the reference's main.rs is written
// against its crate: build a scene with the same API names, Camera::render it (on the GPU, through the C ABI),
// write the Canvas as PPM.
//
//   rtc_example hexagon out.ppm [width]        putting_it_together_hexagon (main.rs:84-146)
//   rtc_example obj <file.obj> out.ppm [width]  an OBJ mesh framed like putting_it_together_cow (main.rs:328-363)
//
// It is an example and a test vehicle (tests/test_gpu_parity.py runs it); the reference's own CLI
// (main.rs:43-81) is out of scope.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "rtc_host.hpp"

using namespace rtc::host;

static World hexagon_world() {
  constexpr double PI = 3.14159265358979323846;
  World world(Light{Tuple::point(0.0, 6.9, -5.0), Color{1.0, 1.0, 0.9}});
  auto corner = [] {
    Shape c = Shape::sphere();
    c.set_transform(translation(0., 0., -1.) * scaling(0.25, 0.25, 0.25));
    return c;
  };
  auto edge = [&] {
    Shape e = Shape::cylinder(0., 1., true);
    e.set_transform(translation(0., 0., -1.) * rotation_y(-PI / 6.) * rotation_z(-PI / 2.) * scaling(0.25, 1., 0.25));
    return e;
  };
  Shape hexagon = Shape::group();
  for (int i = 0; i < 6; ++i) {
    Shape side = Shape::group();
    side.push_shape(corner());
    side.push_shape(edge());
    side.set_transform(rotation_y((double)i * PI / 3.));
    hexagon.push_shape(std::move(side));
  }
  hexagon.set_transform(scaling(2.5, 2.5, 2.5));
  world.objects.push_back(std::move(hexagon));
  return world;
}

int main(int argc, char** argv) {
  try {
    if (argc < 3) {
      std::fprintf(stderr, "usage: %s hexagon <out.ppm> [width] | obj <file.obj> <out.ppm> [width]\n", argv[0]);
      return 2;
    }
    const std::string mode = argv[1];
    int a = 2;
    World world(Light{Tuple::point(0.0, 6.9, -5.0), Color{1.0, 1.0, 0.9}});
    Tuple to = Tuple::point(0.0, 0.0, 0.0);
    if (mode == "hexagon") {
      world = hexagon_world();
    } else if (mode == "obj") {
      if (argc < 4) return 2;
      Shape mesh = Parser::from_obj_file(argv[a++]).obj_to_group();
      mesh.set_transform(translation(0., 3.5, 0.) * scaling(0.5, 0.5, 0.5));
      Material m;
      m.diffuse = 0.7, m.shininess = 300.0, m.reflective = 0.2;
      mesh.set_material(m);
      world.objects.push_back(std::move(mesh));
      to = Tuple::point(0.0, 3.0, 0.0);
    } else {
      std::fprintf(stderr, "unknown scene %s\n", mode.c_str());
      return 2;
    }
    const std::string out = argv[a++];
    const size_t width = a < argc ? (size_t)std::atoi(argv[a]) : 400;
    Camera camera(width, width / 2, 0.785);
    camera.set_transform(view_transform(Tuple::point(8.0, 6.0, -8.0), to, Tuple::vector(0.0, 1.0, 0.0)));
    RenderStats st;
    Canvas canvas = camera.render(world, &st);
    std::ofstream f(out, std::ios::binary);
    if (!f) {
      std::printf("Can't open %s\n", out.c_str());
      return 1;
    }
    f << canvas.to_ppm();
    std::printf("%zux%zu, %llu radiance + %llu shadow rays, kernel %.3f ms\n", canvas.width(), canvas.height(),
                (unsigned long long)st.radiance_rays, (unsigned long long)st.shadow_rays, st.kernel_ms);
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    return 1;
  }
}
