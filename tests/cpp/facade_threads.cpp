// The reference's own threading pattern against the facade (raytracing_project_amd/host/flyscene.hpp), built with
// -fsanitize=thread over tests/cpp/rtx_stub.cpp (a CPU stand-in for the C ABI that makes overlapping calls on one
// rtx_scene a data race).  Driven by tests/test_facade_threads.py; no GPU.
//   facade_threads facade <model.obj>   Flyscene::raytraceScene's body (src/flyscene.cpp:408-430): 16 std::threads calling
//                                       updating_pixels(pixel_data, origin, image_size, 16, i), then 16 threads of per-ray
//                                       queries and batched queries at once; prints "facade ok" when every value is right
//   facade_threads raw                  the same 16 threads straight onto the C ABI, no facade: the negative control --
//                                       ThreadSanitizer must report this one
#include <cstdio>
#include <thread>
#include <vector>

#include "../../raytracing_project_amd/host/flyscene.hpp"

using namespace rtxhost;
extern "C" int rtx_stub_renders(const rtx_scene* s);

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "";
  const int number_threads = 16;                      // src/flyscene.cpp:418
  if (mode == "raw") {
    rtx_scene* s = nullptr;
    float v[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0}; uint32_t idx[3] = {0, 1, 2};
    rtx_scene_create(v, 3, idx, nullptr, 1, nullptr, 0, &s);
    rtx_scene_commit(s, 0);
    std::vector<std::thread> threads;
    for (int i = 0; i < number_threads; ++i)
      threads.emplace_back([s] { float o[3] = {1, 2, 3}, d[3] = {0, 0, 1}, t; int32_t f; for (int k = 0; k < 200; ++k) rtx_trace_closest(s, o, d, 1, &t, &f, nullptr, nullptr, nullptr); });
    for (auto& t : threads) t.join();
    rtx_scene_destroy(s);
    std::printf("raw done\n");
    return 0;
  }
  if (mode != "facade" || argc < 3) { std::fprintf(stderr, "usage: facade_threads facade <model.obj> | raw\n"); return 64; }
  Flyscene fs;
  fs.model_path = argv[2];
  fs.render_parts = 1;
  fs.initialize(96, 64);
  // ---- raytraceScene's body, as the reference writes it (src/flyscene.cpp:402-430) with the [x][y] buffer of F8
  Vector2i image_size(96, 64);
  std::vector<std::vector<Vector3f>> pixel_data(image_size[0]);
  for (int i = 0; i < image_size[0]; ++i) pixel_data[i].resize(image_size[1]);
  Vector3f origin = fs.getCenter();
  std::vector<std::thread> threads;
  for (int i = 0; i < number_threads; i++) {
    threads.emplace_back([&fs, &pixel_data, &origin, &image_size, number_threads, i] { fs.updating_pixels(pixel_data, origin, image_size, number_threads, i); });
  }
  for (auto& t : threads) t.join();
  int bad = 0;
  for (int x = 0; x < image_size[0]; ++x)
    for (int y = 0; y < image_size[1]; ++y) {
      const Vector3f p = pixel_data[x][y];
      if (p[0] != (float)x || p[1] != (float)y || p[2] != (float)(x * 1000 + y)) ++bad;
    }
  if (bad) { std::printf("%d pixels wrong\n", bad); return 1; }
  if (fs.framesRendered() != 1) { std::printf("%d frames rendered by 16 threads, expected 1\n", fs.framesRendered()); return 1; }
  // ---- sixteen threads of per-ray queries, batched queries and a second frame (camera moved) at once
  threads.clear();
  std::vector<int> wrong(number_threads, 0);
  for (int i = 0; i < number_threads; ++i) {
    threads.emplace_back([&fs, &wrong, &pixel_data, &origin, &image_size, number_threads, i] {
      for (int k = 0; k < 50; ++k) {
        const float a = (float)(i + 1), b = (float)(k + 1);
        Ray r(Vector3f(a, b, 0.5f), Vector3f(0.0f, 0.0f, 1.0f));
        auto hit = fs.getIntersections(r);
        if (hit.first.t != std::sqrt(a * a + (b * b + 0.25f)) && std::fabs(hit.first.t - std::sqrt(a * a + b * b + 0.25f)) > 1e-4f) ++wrong[i];
        if (hit.second.material_id != (int)b) ++wrong[i];
        Vector3f c = fs.traceRay(1, r, true);
        if (c[0] != a || c[1] != 1.0f || c[2] != 1.5f) ++wrong[i];
        if (fs.canSeeLight(Vector3f(a + 1.0f, 0, 0), Vector3f(a, 0, 0)).first != true) ++wrong[i];
        if (k % 10 == 0) {
          std::vector<Ray> many;
          for (int q = 0; q < 33; ++q) many.push_back(Ray(Vector3f(a, (float)q, 0.0f), Vector3f(1.0f, 0.0f, 0.0f)));
          auto hs = fs.getIntersections(many);
          auto cs = fs.traceRay(0, many, false);
          for (int q = 0; q < 33; ++q) { if (hs[q].second.material_id != q) ++wrong[i]; if (cs[q][0] != a + 1.0f || cs[q][2] != 0.0f) ++wrong[i]; }
        }
      }
      fs.updating_pixels(pixel_data, origin, image_size, number_threads, i);
    });
  }
  for (auto& t : threads) t.join();
  int total = 0;
  for (int w : wrong) total += w;
  if (total) { std::printf("%d wrong answers\n", total); return 1; }
  if (fs.framesRendered() != 1) { std::printf("the cached frame was rendered again (%d)\n", fs.framesRendered()); return 1; }
  std::printf("facade ok\n");
  return 0;
}
