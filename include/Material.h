// Material.h -- drop-in for PT_sv5_/Material.h: the per-mesh Disney parameter block.
// Layout (104 bytes) and the *non-trivial* default values of the reference are preserved; the
// defaults are what every loadOBJ mesh renders with, since the loader only overrides color and
// emission (Model.cpp:190-191).
#pragma once
#include <cmath>
#include "fovpt.h"
#include "fovpt_vec.h"

static const int MATERIAL_FLAG_NONE = 0;
static const int MATERIAL_FLAG_SHADOW_CATCHER = FOVPT_MATERIAL_FLAG_SHADOW_CATCHER;

struct Material {
    float3 emission = make_float3(1.0f);        //   0
    float3 color = make_float3(1.0f, 0.0f, 0.0f);  //  12
    float3 absorption = make_float3(1.0f);      //  24
    float eta = 1.4f;                           //  36  0 => derive the IOR from `specular`
    float metallic = 0.5f;                      //  40
    float subsurface = 0.0f;                    //  44
    float specular = 1.0f;                      //  48
    float roughness = 1.0f;                     //  52
    float specularTint = 1.0f;                  //  56
    float anisotropic = 0.0f;                   //  60
    float sheen = 0.0f;                         //  64
    float sheenTint = 0.0f;                     //  68
    float clearcoat = 0.0f;                     //  72
    float clearcoatGloss = 1.0f;                //  76
    float transmission = 0.4f;                  //  80
    float bump = 0.0f;                          //  84
    float3 bumpTile = make_float3(1.0f);        //  88
    int flags = 0;                              // 100

    float GetIndexOfRefraction() const
    {
        return eta == 0.0f ? 2.0f / (1.0f - std::sqrt(0.08f * specular)) - 1.0f : eta;
    }
};
static_assert(sizeof(Material) == sizeof(fovpt_material), "Material must stay 104 bytes");
