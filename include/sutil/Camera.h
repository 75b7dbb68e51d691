// sutil/Camera.h -- the part of sutil::Camera (sutil/Camera.h:40-100, Camera.cpp:32-44) the renderer uses.
#pragma once
#include "../fovpt.h"
#include "../fovpt_vec.h"

namespace sutil {
class Camera {
public:
    Camera() : m_eye(make_float3(1.0f)), m_lookat(make_float3(0.0f)), m_up(make_float3(0.0f, 1.0f, 0.0f)), m_fovY(35.0f), m_aspectRatio(1.0f) {}
    Camera(const float3& eye, const float3& lookat, const float3& up, float fovY, float aspectRatio)
        : m_eye(eye), m_lookat(lookat), m_up(up), m_fovY(fovY), m_aspectRatio(aspectRatio) {}
    const float3& eye() const { return m_eye; }
    void setEye(const float3& v) { m_eye = v; }
    const float3& lookat() const { return m_lookat; }
    void setLookat(const float3& v) { m_lookat = v; }
    const float3& up() const { return m_up; }
    void setUp(const float3& v) { m_up = v; }
    const float& fovY() const { return m_fovY; }
    void setFovY(const float& v) { m_fovY = v; }
    const float& aspectRatio() const { return m_aspectRatio; }
    void setAspectRatio(const float& v) { m_aspectRatio = v; }
    // U, V, W are orthogonal but not normalised: |W| is the focal length
    void UVWFrame(float3& U, float3& V, float3& W) const
    {
        fovpt_camera_uvw((const fovpt_float3*)&m_eye, (const fovpt_float3*)&m_lookat, (const fovpt_float3*)&m_up,
                         m_fovY, m_aspectRatio, (fovpt_float3*)&U, (fovpt_float3*)&V, (fovpt_float3*)&W);
    }
private:
    float3 m_eye, m_lookat, m_up;
    float m_fovY, m_aspectRatio;
};
}  // namespace sutil
