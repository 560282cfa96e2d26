// Host-side Material: the fields and constructor defaults of the reference's Material (src/Material.hpp:157-167,245-257).
// The BSDF itself (sample/eval/pdf/fresnel) runs on the GPU (csrc/mcpt_device.h); the host only describes materials.
#pragma once
#include "Vector.hpp"

enum MaterialType { SMOOTH_CONDUCTOR, ROUGH_CONDUCTOR, SMOOTH_DIELECTRIC, ROUGH_DIELECTRIC };  // Material.hpp:13-18

class Material {
  public:
    MaterialType m_type;
    Vector3f m_emission;
    float iorA, iorB;
    bool textured = false;  // the reference leaves this uninitialised (Material.hpp:164); false is what its renders show
    bool isDirac;
    float roughness;
    Vector3f base_reflectance;

    explicit Material(MaterialType t = ROUGH_CONDUCTOR, Vector3f e = Vector3f(0, 0, 0)) {
        m_type = t;
        m_emission = e;
        isDirac = (t == SMOOTH_CONDUCTOR || t == SMOOTH_DIELECTRIC);
        iorA = 1.74;
        iorB = 0.1f;
        roughness = (t == ROUGH_DIELECTRIC) ? 0.2f : 1.f;
        base_reflectance = Vector3f(0, 0, 0);
    }
    MaterialType getType() const { return m_type; }
    Vector3f getEmission() const { return m_emission; }
    bool hasEmission() const { return m_emission.norm() > 1e-4f; }  // Material.hpp:262
};
