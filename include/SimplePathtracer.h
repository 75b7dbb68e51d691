// SimplePathtracer.h -- drop-in for PT_sv5_/SimplePathtracer.h: class SampleRenderer with the same
// public interface (ctor, render x2, resize, downloadPixels, setCamera, setProbe; public
// launchParams and stream), implemented over the C ABI of libfovpt (include/fovpt.h) instead of
// OptiX.  Header-only; link with -lfovpt.  Errors become std::runtime_error like the reference's
// sutil::Exception (sutil/Exception.h:245).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "LaunchParams.h"
#include "Model.h"
#include "sutil/Camera.h"

class SampleRenderer {
public:
    // performs all setup: device context + scene upload + LBVH build (SimplePathtracer.cpp:42-75)
    explicit SampleRenderer(const Model* model, int device = 0) : model(model)
    {
        if (fovpt_create(&ctx, device) != FOVPT_OK) throw std::runtime_error(fovpt_last_error(nullptr));
        std::vector<fovpt_mesh_desc> md(model->meshes.size());
        for (size_t i = 0; i < md.size(); i++) {
            const TriangleMesh& m = *model->meshes[i];
            md[i].vertex = m.vertex.empty() ? nullptr : &m.vertex[0].x;
            md[i].normal = m.normal.empty() ? nullptr : &m.normal[0].x;
            md[i].texcoord = m.texcoord.empty() ? nullptr : &m.texcoord[0].x;
            md[i].index = m.index.empty() ? nullptr : &m.index[0].x;
            md[i].num_vertices = (uint32_t)m.vertex.size();
            md[i].num_triangles = (uint32_t)m.index.size();
            md[i].texture_id = m.diffuseTextureID;
            static_assert(sizeof(Material) == sizeof(fovpt_material), "");
            md[i].material = *reinterpret_cast<const fovpt_material*>(&m.material);
        }
        std::vector<fovpt_texture_desc> td(model->textures.size());
        for (size_t i = 0; i < td.size(); i++) {
            td[i].pixel = model->textures[i]->pixel;
            td[i].width = model->textures[i]->resolution.x;
            td[i].height = model->textures[i]->resolution.y;
        }
        uint64_t trav = 0;
        check(fovpt_set_scene(ctx, md.data(), (int)md.size(), td.data(), (int)td.size(), &trav));
        launchParams.traversable = trav;
        stream = fovpt_stream(ctx);
    }
    ~SampleRenderer() { fovpt_destroy(ctx); }
    SampleRenderer(const SampleRenderer&) = delete;
    SampleRenderer& operator=(const SampleRenderer&) = delete;

    // one frame: the three foveation passes (or FOV_OFF), then a device sync (SimplePathtracer.cpp:77-214)
    void render()
    {
        check(fovpt_render(ctx, reinterpret_cast<fovpt_launch_params*>(&launchParams)));
        check(fovpt_synchronize(ctx));
    }
    // render into a caller-owned target: anything with `uint32_t* map()` / `void unmap()`, i.e. the shape
    // of sutil::CUDAOutputBuffer<uint32_t>.  Like the reference this repoints frame_buffer and leaves it (:218-219).
    template <typename Target>
    void render(Target& renderTarget)
    {
        uint32_t* result_buffer_data = renderTarget.map();
        launchParams.frame.frame_buffer = (uchar4*)result_buffer_data;
        render();
        renderTarget.unmap();
    }
    // the literal north-star overload: render with caller-held parameters
    void render(LaunchParams& params)
    {
        check(fovpt_render(ctx, reinterpret_cast<fovpt_launch_params*>(&params)));
        check(fovpt_synchronize(ctx));
    }
    void resize(const int2& newSize)
    {
        if (newSize.x == 0 || newSize.y == 0) return;
        fovpt_frame_ptrs p;
        check(fovpt_resize(ctx, newSize.x, newSize.y, &p));
        own_frame = p.frame_buffer;
        launchParams.frame.size = newSize;
        launchParams.frame.frame_buffer = (uchar4*)p.frame_buffer;
        launchParams.frame.accum_buffer = (float4*)p.accum_buffer;
        launchParams.frame.normal_buffer = (float4*)p.normal_buffer;
        launchParams.frame.color_buffer = (float4*)p.color_buffer;
        launchParams.frame.albedo_buffer = (float4*)p.albedo_buffer;
    }
    // always the renderer's own frame buffer (SimplePathtracer.cpp:276-280)
    void downloadPixels(uint32_t h_pixels[])
    {
        check(fovpt_download(ctx, own_frame, h_pixels, sizeof(uint32_t) * (size_t)launchParams.frame.size.x * (size_t)launchParams.frame.size.y));
    }
    void setCamera(const sutil::Camera& camera)
    {
        lastSetCamera = camera;
        lastSetCamera.setAspectRatio(launchParams.frame.size.x / float(launchParams.frame.size.y));
        lastSetCamera.UVWFrame(launchParams.camera.U, launchParams.camera.V, launchParams.camera.W);
        launchParams.camera.eye = lastSetCamera.eye();
    }
    void setProbe(const ProbeData& probe)
    {
        if (!probe.valid) throw std::runtime_error("Probe Data is not valid");       // Probe.h:104-105
        check(fovpt_set_probe(ctx, probe.width, probe.height, (const fovpt_float4*)probe.data, probe.pdfValuesX, probe.cdfValuesX,
                              probe.pdfValuesY, probe.cdfValuesY, (const fovpt_float3*)&probe.offset, &launchParams.probe));
    }
    // ---- multi-GPU (new with this library; the reference is single-GPU): one SampleRenderer per GPU / process, rank and
    // world in fovpt_config, the framebuffer gathered over RCCL on the library's stream (include/fovpt.h, fovpt_comm_*)
    void renderAsync() { check(fovpt_render(ctx, reinterpret_cast<fovpt_launch_params*>(&launchParams))); }   // render() without the sync
    // frames issued with renderAsync() run up to two at a time (fovpt_config.frames_in_flight: 2 = throughput, the default;
    // 1 = lowest latency per frame); they complete in order on `stream`
    void setFramesInFlight(int n) { fovpt_config c = config(); c.frames_in_flight = n; setConfig(c); }
    void commInit(const void* uniqueId, int rank, int world) { check(fovpt_comm_init(ctx, uniqueId, rank, world)); }
    // gathers the frame just rendered (the renderer's current frame_buffer) onto `root`; fullFrame: device memory, root only
    void gatherFrame(int root, uint32_t* fullFrame)
    {
        check(fovpt_gather_frame(ctx, reinterpret_cast<const fovpt_launch_params*>(&launchParams), root,
                                 (const uint32_t*)launchParams.frame.frame_buffer, fullFrame));
    }
    // the reference's compile-time switches (FOV_ON/OFF, radii, spp, depth) as run-time settings
    fovpt_config config() const { fovpt_config c; fovpt_get_config(ctx, &c); return c; }
    void setConfig(const fovpt_config& c) { check(fovpt_set_config(ctx, &c)); }

    LaunchParams launchParams;       // public and caller-mutated, as in the reference (SimplePathtracer.h:146)
    void* stream = nullptr;          // hipStream_t
    sutil::Camera lastSetCamera;
    const Model* model;

private:
    void check(int rc) const
    {
        if (rc != FOVPT_OK) throw std::runtime_error(std::string("libfovpt: ") + fovpt_last_error(ctx));
    }
    fovpt_ctx* ctx = nullptr;
    uint32_t* own_frame = nullptr;
};
