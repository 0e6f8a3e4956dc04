// rtm_internal.h -- shared between the host front end and the HIP back end (not installed).
#pragma once
#include <cstdint>
#include <string>

// Records `msg` as the thread's last error (rtm_last_error) and returns `code`.
int rtm_internal_fail(int code, const std::string& msg);

// ---- device-resident scene ---------------------------------------------------------------------
// Layout in HBM (DESIGN.md "Data layout"): analytic primitives and tables are small AoS records read
// with wave-uniform (scalar) loads; triangles are per-triangle records because a lane gathers whole
// triangles at its own index; the grid is an EXCLUSIVE prefix array with n^3+1 entries, so a cell's
// [start,end) is one 8-byte load.
struct RtmDevModel {
    double bmin[3];   // center - half                (box3D.rs:55-57)
    double bmax[3];   // center + half                (box3D.rs:59-61)
    double cell[3];   // (bmax - bmin) / n            (model.rs:243-245)
    double hc[3];     // half / n                     (model.rs:232)
    double hc2[3];    // hc * 2.0                     (model.rs:235-236)
    uint32_t n;       // cells per side
    uint32_t n_tris;
    uint32_t mat, obj;
    const uint32_t* cell_start;  // n^3 + 1, exclusive prefix (from the inclusive offset_array)
    const uint32_t* tri_idx;
    const double* tri_pos;       // n_tris x 9: v0 xyz, (v1-v0) xyz, (v2-v0) xyz  -- edges of model.rs:46-47 / :20-21
    const double* tri_nrm;       // n_tris x 9
    const uint32_t* occupancy;   // n^3 bits: cell non-empty
    const uint32_t* occupancy7;  // seven bitmaps back to back: arrival directions -x,+x,-y,+y,-z,+z (bit = the cell has an entry that was
                                 // not already tested in the neighbour the step came from), then the plain occupancy (entry cell)
    const uint32_t* cell_info;   // n^3 x 8: list start, list end, then six masks -- which of the (<= 32) entries are NOT also listed in the
                                 // neighbour cell a DDA step along -x,+x,-y,+y,-z,+z came from (rt_front.h)
    const double* cell_rec;      // n_tri_idx x 10, in list order: the tri_pos record of entry k, then its triangle index (as integer bits)
};

struct RtmDevScene {
    double cam_pos[3], cam_fwd[3], cam_up[3], cam_right[3], cam_zoom;
    uint32_t n_objects, n_spheres, n_planes, n_models, n_lights, n_materials;
    const double* spheres;    // n_spheres x 4: cx cy cz r
    const uint32_t* sph_ids;  // n_spheres x 2: mat, obj
    const double* planes;     // n_planes x 4: nx ny nz d
    const uint32_t* pl_ids;   // n_planes x 2: mat, obj
    const RtmDevModel* models;
    const double* lights;     // n_lights x 8: px py pz r g b radius pad
    const double* materials;  // n_materials x 4: r g b type(0.0|1.0)
};
