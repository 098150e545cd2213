// dw_bufg.h -- how the octet kernels receive the buffer table (DwBuffers, 24 pointers).
//
// By value the table is 48 scalar registers for the whole launch; the step kernel has no room for them, the compiler parks scalars
// in the lanes of a vector register and fetches them back with v_readlane (a vector instruction each, 1 330 in the listing).  Read
// from the parameter block at the place of use a pointer costs a scalar load and its latency instead -- measured: slower at 16384
// envs when done for ALL pointers.  Hence the split: the ten pointers of the physics and the item loops travel
// by value (DwHot), the fourteen that the post phase touches once per step are read from the parameter block through a mirror of
// DwBuffers with address-space-1 member types, so that those accesses are global, not FLAT, instructions.
#pragma once

#include <stddef.h>

#include "../../include/dyros_walk.h"

#if defined(__HIP_DEVICE_COMPILE__)          // (the device pass only: on the host pass of hipcc, as under g++, it is a plain pointer)
#define DW_GPTR __attribute__((address_space(1)))
#else
#define DW_GPTR
#endif

struct DwBuffersG {
    float   DW_GPTR *root_states;
    float   DW_GPTR *dof_state;
    float   DW_GPTR *contact_forces;
    float   DW_GPTR *mass_scale;
    float   DW_GPTR *dof_damping;
    float   DW_GPTR *dof_armature;
    float   DW_GPTR *friction_scale;
    float   DW_GPTR *total_mass;
    float   DW_GPTR *env_origins;
    float   DW_GPTR *obs_buf;
    float   DW_GPTR *rew_buf;
    int64_t DW_GPTR *reset_buf;
    int64_t DW_GPTR *progress_buf;
    int64_t DW_GPTR *timeout_buf;
    int64_t DW_GPTR *randomize_buf;
    float   DW_GPTR *stacked_rewards;
    float   DW_GPTR *env_state;
    float   DW_GPTR *obs_history;
    float   DW_GPTR *action_history;
    int64_t DW_GPTR *gate_acc;
    int16_t DW_GPTR *height_samples;
    float   DW_GPTR *terrain_origins;
    int64_t DW_GPTR *terrain_levels;
    int64_t DW_GPTR *terrain_types;
};
static_assert(sizeof(DwBuffersG) == sizeof(DwBuffers), "DwBuffersG mirrors DwBuffers");
#define DW_BUFG_SAME(f) static_assert(offsetof(DwBuffersG, f) == offsetof(DwBuffers, f), "DwBuffersG mirrors DwBuffers: " #f)
DW_BUFG_SAME(root_states); DW_BUFG_SAME(dof_state); DW_BUFG_SAME(contact_forces); DW_BUFG_SAME(mass_scale); DW_BUFG_SAME(dof_damping);
DW_BUFG_SAME(dof_armature); DW_BUFG_SAME(friction_scale); DW_BUFG_SAME(total_mass); DW_BUFG_SAME(env_origins); DW_BUFG_SAME(obs_buf);
DW_BUFG_SAME(rew_buf); DW_BUFG_SAME(reset_buf); DW_BUFG_SAME(progress_buf); DW_BUFG_SAME(timeout_buf); DW_BUFG_SAME(randomize_buf);
DW_BUFG_SAME(stacked_rewards); DW_BUFG_SAME(env_state); DW_BUFG_SAME(obs_history); DW_BUFG_SAME(action_history); DW_BUFG_SAME(gate_acc);
DW_BUFG_SAME(height_samples); DW_BUFG_SAME(terrain_origins); DW_BUFG_SAME(terrain_levels); DW_BUFG_SAME(terrain_types);
#undef DW_BUFG_SAME

// the pointers that travel by value, and the table as the kernels' code sees it: B.root_states ... as before, OQ_COLD(rew_buf) ... for
// the rest
struct DwHot { float *root_states, *dof_state, *contact_forces, *mass_scale, *dof_damping, *dof_armature, *obs_buf, *env_state, *obs_history, *action_history; };
struct OBuf {
    float *root_states, *dof_state, *contact_forces, *mass_scale, *dof_damping, *dof_armature, *obs_buf, *env_state, *obs_history, *action_history;
    const DwBuffersG *cold;          // the whole table in the parameter block (device memory), global-pointer view
    const DwBuffers  *all;           // the same bytes as the C-ABI struct (for code shared with the other kernel generations)
};
DQ_HD OBuf make_obuf(const DwHot &h, const DwBuffers *table) {
    OBuf b;
    b.root_states = h.root_states; b.dof_state = h.dof_state; b.contact_forces = h.contact_forces; b.mass_scale = h.mass_scale;
    b.dof_damping = h.dof_damping; b.dof_armature = h.dof_armature; b.obs_buf = h.obs_buf; b.env_state = h.env_state;
    b.obs_history = h.obs_history; b.action_history = h.action_history;
    b.cold = reinterpret_cast<const DwBuffersG *>(table); b.all = table;
    return b;
}
inline DwHot make_hot(const DwBuffers &t) {
    return DwHot{t.root_states, t.dof_state, t.contact_forces, t.mass_scale, t.dof_damping, t.dof_armature, t.obs_buf, t.env_state, t.obs_history, t.action_history};
}
#define OQ_COLD(f) (B.cold->f)
