// vmorph/parameters.hpp -- C++ host mirror of Algorithm/parameters.h (reference
// repository): same type and field names, CUDA's int2/int4 and OpenCV dropped.
#ifndef VMORPH_PARAMETERS_HPP
#define VMORPH_PARAMETERS_HPP

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vmorph.h"

namespace vmorph {

struct int2 { int x, y; };
struct int4 { int x, y, z, w; };

// enum BoundaryCondition, parameters.h:9-14
enum BoundaryCondition { BCOND_NONE = VM_BCOND_NONE, BCOND_CORNER = VM_BCOND_CORNER, BCOND_BORDER = VM_BCOND_BORDER };

// parameters.h:16-27
struct Connect { int2 li; int2 ri; };
struct Conp { int4 p; float weight; };

// struct Parameters, parameters.h:29-52; defaults of UI/MdiEditor.cpp:131-140
struct Parameters {
    int frame0 = 0, frame1 = 0;
    int2 range0{0, 0}, range1{0, 0};
    int total_frame = 1;

    float w_ui = 1e5f, w_tps = 0.05f, w_ssim = 100.0f, w_temp = 10.0f;
    float ssim_clamp = 0.0f;
    float eps = 0.01f;

    int max_iter = 1000;
    int start_res = 8;
    float max_iter_drop_factor = 2.0f;

    BoundaryCondition bcond = BCOND_NONE;

    std::vector<std::vector<Conp>> lp;
    std::vector<std::vector<Conp>> rp;
    std::vector<std::vector<Connect>> cnt;

    int2 ActIndex_l{-1, -1}, ActIndex_r{-1, -1};
    bool verbose = false;

    // the resolution of lp/rp/cnt that morph.cu:354-366 performs for page `conz`
    std::vector<vm_constraint> constraints(int conz = 0) const
    {
        std::vector<vm_constraint> out;
        for (const auto &row : cnt)
            for (const Connect &c : row) {
                const Conp &l = lp.at(c.li.x).at(c.li.y);
                const Conp &r = rp.at(c.ri.x).at(c.ri.y);
                if (l.p.z != conz) continue;
                out.push_back(vm_constraint{(float)l.p.x, (float)l.p.y, (float)r.p.x, (float)r.p.y,
                                            std::min(l.weight, r.weight)});
            }
        return out;
    }
};

// struct KernParameters, parameters.h:54-72
struct KernParameters : vm_kern_params {
    KernParameters() : vm_kern_params{} {}
    KernParameters(const Parameters &p)
    {
        w_temp = p.w_temp; w_ui = p.w_ui; w_tps = p.w_tps; w_ssim = p.w_ssim;
        ssim_clamp = p.ssim_clamp; eps = p.eps; bcond = p.bcond;
    }
};

// rod::check_cuda_error throws std::runtime_error (include/util/error.cpp:9-23):
// the facade re-raises the C-ABI's error codes the same way
inline void check(int rc)
{
    if (rc != VM_OK)
        throw std::runtime_error(std::string("vmorph: ") + vm_last_error());
}

} // namespace vmorph
#endif
