// Reference-side adapter, part 2: the GPU sibling of CPURenderer (rendering/CPURenderer.h), to be
// dropped into the reference tree as src/rendering/HIPRenderer.{h,cpp} and linked with -lmrt_hip.
// main.cpp:36 then chooses:  ImageRenderer* renderer = useGpu ? new HIPRenderer : new CPURenderer;
#pragma once
#include "ImageRenderer.h"
#include "flatten_ref.h"

#include <mrt.h>

#include <vector>

class HIPRenderer : public ImageRenderer
{
  public:
    void init( std::string& imageName, glm::vec2 resolution, Scene* scene ) override;
    void destroy() override;
    void run() override;

    // the literals of CPURenderer.cpp:15,18,42 as parameters
    int spp_x = 32, spp_y = 32, maxDepth = 5;
    unsigned seed = 0;            // 0: time(NULL), as the reference seeds
    std::vector<int> devices { 0 }; // more than one: the samples are divided among them

  private:
    RenderImage* m_renderImage = nullptr;
    KdTree* m_kdTree           = nullptr;
    MrtFlat m_flat;
    std::vector<mrt_scene*> m_dev;
};
