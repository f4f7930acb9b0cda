// Wall time of Flyscene::raytraceScene through the C++ facade (render + read-back + result.ppm), the call a user of the
// reference makes (src/main.cpp:63-65).  usage: facade_time <model.obj> <width> <height> [north_star]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../raytracing_project_amd/host/flyscene.hpp"

using namespace rtxhost;

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s model.obj width height [north_star]\n", argv[0]); return 2; }
  try {
    Flyscene fs;
    fs.model_path = argv[1];
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    auto ti0 = std::chrono::steady_clock::now();
    fs.initialize(W, H);
    std::printf("initialize (OBJ + MTL import, hierarchy, upload): %.1f ms, %d faces\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count(), fs.getNumberOfFaces());
    if (argc > 4 && std::strcmp(argv[4], "north_star") == 0) {       // primary + shadow only (BASELINE.json's options)
      fs.options.max_bounces = 0; fs.options.refraction = 0; fs.options.gi_rays = 0;
    }
    fs.render_parts = 1;
    {   // the parts of the call, each on a view of its own
      fs.getCamera()->view[11] -= 0.01f;
      auto t0 = std::chrono::steady_clock::now();
      fs.render_frame(W, H);
      auto t1 = std::chrono::steady_clock::now();
      writePPMImageRowMajor("result.ppm", fs.frame_data(), W, H);
      auto t2 = std::chrono::steady_clock::now();
      std::printf("first frame: render + read-back %.1f ms, result.ppm %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                  std::chrono::duration<double, std::milli>(t2 - t1).count());
      fs.getCamera()->view[11] -= 0.01f;
      t0 = std::chrono::steady_clock::now();
      fs.render_frame(W, H);
      t1 = std::chrono::steady_clock::now();
      writePPMImageRowMajor("result.ppm", fs.frame_data(), W, H);
      t2 = std::chrono::steady_clock::now();
      std::printf("second frame: render + read-back %.1f ms, result.ppm %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                  std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    for (int k = 0; k < 3; ++k) {
      fs.getCamera()->view[11] -= 0.01f;                               // another view: the cached frame does not count
      auto t0 = std::chrono::steady_clock::now();
      fs.raytraceScene(W, H);
      auto t1 = std::chrono::steady_clock::now();
      std::printf("raytraceScene %dx%d call %d: %.1f ms\n", W, H, k, std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
