// rtm_internal.h -- shared between the host front end and the HIP back end (not installed).
#pragma once
#include <cstdint>
#include <string>

// Records `msg` as the thread's last error (rtm_last_error) and returns `code`.
int rtm_internal_fail(int code, const std::string& msg);
// The device rtm_init selected for this process, or -1 before rtm_init / after rtm_shutdown (render.hip).
int rtm_internal_device();

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
    uint32_t cell_base;          // first cell of this model in the scene-wide arrays of the ray-front path (a multiple of 32)
    uint32_t reserved;
};

struct RtmDevScene {
    double cam_pos[3], cam_fwd[3], cam_up[3], cam_right[3], cam_zoom;
    uint32_t n_objects, n_spheres, n_planes, n_models, n_lights, n_materials;
    const double* spheres;    // n_spheres x 4: cx cy cz r
    const uint32_t* sph_ids;  // n_spheres x 2: mat, obj
    const double* planes;     // n_planes x 4: nx ny nz d
    const uint32_t* pl_ids;   // n_planes x 2: mat, obj
    const float* sph_screen;  // the spheres again for the shade kernel's packed-f32 screen (rt_front.h), two per record: cx0 cx1 cy0 cy1 cz0 cz1
                              // r0^2 r1^2 rounded to f32, padded with zeros to a multiple of 32 spheres
    float sph_lc, sph_r2max;  // max |centre component| and max r^2 over the spheres (rounded up)
    uint32_t sph_screen_ok, sph_reserved;  // 0: some sphere is outside the range the screen's error bound holds for
    const RtmDevModel* models;
    const double* lights;     // n_lights x 8: px py pz r g b radius pad
    const double* materials;  // n_materials x 4: r g b type(0.0|1.0)
    // ---- the ray-front path's grids, over the cells of all models (model m's cell c is scene cell models[m].cell_base + c):
    const uint32_t* g_cell_info;  // cells x 8: list start, list end (positions in g_cell_rec), then six masks -- which of the (<= 32)
                                  // entries are NOT also listed in the neighbour cell a DDA step along -x,+x,-y,+y,-z,+z came from;
                                  // cells with more than 32 entries ("long cells") hold their frame index in word 2 instead
    const double* g_cell_rec;     // entries x 10, in list order: v0, v1-v0, v2-v0 of the entry's triangle, then its index in the model
    const uint32_t* g_occ7;       // seven bitmaps, interleaved in groups of eight words per 32 cells (g_occ_dir_words groups): arrival directions
                                  // -x,+x,-y,+y,-z,+z (bit = the cell has an entry that was not already tested in the neighbour the step
                                  // came from), the plain occupancy, one spare word
    uint32_t g_occ_dir_words, g_reserved;
    const double* g_cell_frame;   // long cells x 8: centre xyz of the region that bounds the cell's triangles, quanta per unit length xyz,
                                  // (index of the cell's first box in g_cell_box8) - (list position of its first entry) as integer bits, 1 spare
    const uint32_t* g_cell_box8;  // two words per entry of a long cell, the cell's entries in list order from a 16-byte boundary: the
                                  // triangle's bounding box in the region's 8-bit local coordinates, rounded outwards:
                                  // word 0 = x0 | y0<<8 | z0<<16 | x1<<24, word 1 = y1 | z1<<8
};
