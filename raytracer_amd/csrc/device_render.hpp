// placeholder: filled in with the wavefront path-tracing kernels
#pragma once
#include "device_trace.hpp"
struct DMaterial { float albedo[3]; float emission[3]; float ior, roughness, metalness; int32_t type, texture; };
struct DTexture { int32_t kind; uint32_t w, h; uint32_t pad; uint64_t offset; float even[3], odd[3]; DXf xf; };
struct DLight { int32_t kind; float color[3]; int32_t shape_kind; uint32_t xf; float p[9]; float n[9]; float sphere[4]; };
struct DShade { int dummy; };
struct RenderWorkspace { int dummy; };
