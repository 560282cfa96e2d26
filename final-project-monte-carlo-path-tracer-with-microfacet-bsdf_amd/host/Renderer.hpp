// src/Renderer.hpp:14-23 restated: same members, same defaults.
#pragma once
#include <string>

#include "Scene.hpp"

class Renderer {
  public:
    void Render(const Scene &scene);
    void setSpp(int s) { spp = s; }
    std::string path = "./output.png";

  private:
    int spp = 2048;
};
