// dw_handle.h -- what a DwHandle (include/dyros_walk.h) points to; shared by the translation units that implement C-ABI entry
// points (dw_hip.hip, dw_amp.hip).
#pragma once

#include "dw_params.h"

namespace dwq { struct QuadModel; }

struct DwHandle {
    DwConfig        cfg;
    dw::TaskParams  params;
    dw::DevModel   *d_model;
    dwq::QuadModel *d_qmodel;
    dw::DevParams  *d_params;
    int             pipeline;       // 3 = the octet kernels (8 lanes per env), one launch per policy step
    int             hex;            // this handle's launches use the hex instantiation (16 lanes per env): N <= 4096, or DwConfig.debug_wave_build = 3
    float          *d_mocap;
    float          *d_sc_park;      // PhysParams::sc_park
    int16_t        *d_hmax;         // height field: the coarse bound table built at dw_bind (PhysParams::hmax)
    unsigned long long *d_lvl_acc;  // TaskParams::terrain_lvl_acc (terrain curriculum only)
    float           reach;          // dw_physics.h model_reach() of the model, computed at dw_create
    DwBuffers       buf;
    int             bound;
    int             has_task;
    int             device;
    long long       next_step;      // host-supplied step index the next dw_step is expected to carry (dw_hip.hip launch_step; dw_create: -1 = none yet)
    // the one-launch sibling-task step (dw_amp_step): its two argument tables as they were last copied to the device, and that copy.  The
    // kernel reads them through a pointer (scalar loads): as by-value kernel arguments next to the octet substep they were demoted to
    // scratch, 600 B per lane read field by field.
    void           *d_amp_args;
    unsigned char   amp_args_host[1024];
    int             amp_args_valid;
};
