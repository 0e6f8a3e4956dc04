// The reference's whole program, main.rs:134-192, on the MI355X back end: scene file -> bitmap.
//
//   raytracer_mi355x <scene> [-o output.bmp] [-s 20] [--gamma-correction]        (main.rs:17-31, clap)
//
// Host code only (the device work is behind rtm_scene_upload / rtm_render_rgb8).  The bitmap writer follows the BMP
// encoder of the `image` crate the reference saves with (main.rs:190; image 0.25.1, Cargo.lock): BITMAPFILEHEADER +
// BITMAPINFOHEADER, 24 bits per pixel, rows bottom-up, BGR, rows padded to 4 bytes, resolution fields 0.  That
// crate's source is not part of /root/reference, so byte equality with its files is restated from the format, not
// pinned by a reference fixture.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rtm_internal.h"

namespace {

void put_u16(std::vector<uint8_t>& b, uint32_t v) {
    b.push_back((uint8_t)(v & 255u));
    b.push_back((uint8_t)((v >> 8) & 255u));
}
void put_u32(std::vector<uint8_t>& b, uint32_t v) {
    put_u16(b, v & 0xFFFFu);
    put_u16(b, v >> 16);
}

// Rust's `{:?}` of a Duration (main.rs:188): the largest unit with a non-zero integer part, fraction without
// trailing zeros.
std::string rust_duration(double seconds) {
    if (seconds < 0) seconds = 0;
    const unsigned long long ns = (unsigned long long)(seconds * 1e9 + 0.5);
    unsigned long long whole, frac;
    int digits;
    const char* unit;
    if (ns >= 1000000000ull) { whole = ns / 1000000000ull; frac = ns % 1000000000ull; digits = 9; unit = "s"; }
    else if (ns >= 1000000ull) { whole = ns / 1000000ull; frac = ns % 1000000ull; digits = 6; unit = "ms"; }
    else if (ns >= 1000ull) { whole = ns / 1000ull; frac = ns % 1000ull; digits = 3; unit = "\xC2\xB5s"; }
    else { whole = ns; frac = 0; digits = 0; unit = "ns"; }
    std::string out = std::to_string(whole);
    if (frac) {
        char buf[16];
        std::snprintf(buf, sizeof buf, "%0*llu", digits, frac);
        std::string f(buf);
        while (!f.empty() && f.back() == '0') f.pop_back();
        out += "." + f;
    }
    return out + unit;
}

const char* kUsage =
    "Usage: raytracer_mi355x [OPTIONS] <SCENE>\n\n"
    "Arguments:\n  <SCENE>  the input path to the scene file\n\n"
    "Options:\n"
    "  -o, --output <OUTPUT>            the path where is saved the rendered image as bitmap [default: output.bmp]\n"
    "  -s, --sample-rate <SAMPLE_RATE>  the number of ray shooted per pixel [default: 20]\n"
    "      --gamma-correction           apply gamma correction to the final image\n"
    "      --seed <SEED>                seed of the counter-based random numbers [default: 0] (this back end only)\n"
    "      --base-dir <DIR>             directory model paths are resolved against [default: .] (this back end only)\n"
    "      --synth-normals              geometric normals for OBJ faces without vn (this back end only)\n"
    "  -h, --help                       Print help\n"
    "  -V, --version                    Print version\n";

}  // namespace

extern "C" {

int rtm_write_bmp(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8) {
    if (!path || (!rgb8 && width && height)) return rtm_internal_fail(RTM_ERR_INVALID, "rtm_write_bmp: null argument");
    const uint64_t row = (uint64_t)width * 3u;
    const uint32_t pad = (uint32_t)((4u - row % 4u) % 4u);
    const uint64_t image_size = (row + pad) * height;
    if (width > 0x7FFFFFFFu || height > 0x7FFFFFFFu || image_size + 54u > 0xFFFFFFFFull)
        return rtm_internal_fail(RTM_ERR_INVALID, "rtm_write_bmp: image too large for the BMP format");
    std::vector<uint8_t> b;
    b.reserve((size_t)image_size + 54u);
    b.push_back('B'); b.push_back('M');
    put_u32(b, (uint32_t)(54u + image_size));  // file size
    put_u16(b, 0); put_u16(b, 0);              // reserved
    put_u32(b, 54u);                           // offset of the pixel data
    put_u32(b, 40u);                           // BITMAPINFOHEADER
    put_u32(b, width); put_u32(b, height);     // positive height: bottom-up
    put_u16(b, 1); put_u16(b, 24);
    put_u32(b, 0);                             // BI_RGB
    put_u32(b, (uint32_t)image_size);
    put_u32(b, 0); put_u32(b, 0);              // pixels per metre
    put_u32(b, 0); put_u32(b, 0);              // palette colours, important colours
    for (uint32_t y = height; y-- > 0;) {
        const uint8_t* src = rgb8 + (size_t)y * row;
        for (uint32_t x = 0; x < width; ++x) {
            b.push_back(src[x * 3 + 2]); b.push_back(src[x * 3 + 1]); b.push_back(src[x * 3]);
        }
        for (uint32_t k = 0; k < pad; ++k) b.push_back(0);
    }
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return rtm_internal_fail(RTM_ERR_IO, std::string("cannot create ") + path);
    const size_t n = std::fwrite(b.data(), 1, b.size(), f);
    const int rc = std::fclose(f);
    if (n != b.size() || rc != 0) return rtm_internal_fail(RTM_ERR_IO, std::string("cannot write ") + path);
    return RTM_OK;
}

int rtm_cli_main(int argc, const char* const* argv) {
    std::string scene, output = "output.bmp", base_dir = ".";
    long long spp = 20;
    unsigned long long seed = 0;
    bool gamma = false, synth = false;
    auto fail_usage = [&](const std::string& msg) {
        std::fprintf(stderr, "error: %s\n\n%s", msg.c_str(), kUsage);
        return 2;  // clap's exit status for a usage error
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i] ? argv[i] : "";
        auto value = [&](std::string& into) {
            if (i + 1 >= argc) return false;
            into = argv[++i];
            return true;
        };
        std::string v;
        if (a == "-h" || a == "--help") { std::fputs(kUsage, stdout); std::fflush(stdout); return 0; }
        if (a == "-V" || a == "--version") { std::printf("raytracer_mi355x %s\n", rtm_version()); std::fflush(stdout); return 0; }
        if (a == "-o" || a == "--output") { if (!value(output)) return fail_usage("a value is required for '--output <OUTPUT>'"); continue; }
        if (a.rfind("--output=", 0) == 0) { output = a.substr(9); continue; }
        if (a == "-s" || a == "--sample-rate" || a.rfind("--sample-rate=", 0) == 0) {
            if (a.rfind("--sample-rate=", 0) == 0) v = a.substr(14);
            else if (!value(v)) return fail_usage("a value is required for '--sample-rate <SAMPLE_RATE>'");
            char* end = nullptr;
            spp = std::strtoll(v.c_str(), &end, 10);
            if (v.empty() || *end || spp < 1 || spp > 0xFFFFFFFFll) return fail_usage("invalid value '" + v + "' for '--sample-rate <SAMPLE_RATE>'");
            continue;
        }
        if (a == "--gamma-correction") { gamma = true; continue; }
        if (a == "--synth-normals") { synth = true; continue; }
        if (a == "--seed") {
            if (!value(v)) return fail_usage("a value is required for '--seed <SEED>'");
            char* end = nullptr;
            seed = std::strtoull(v.c_str(), &end, 10);
            if (v.empty() || *end) return fail_usage("invalid value '" + v + "' for '--seed <SEED>'");
            continue;
        }
        if (a == "--base-dir") { if (!value(base_dir)) return fail_usage("a value is required for '--base-dir <DIR>'"); continue; }
        if (a.size() > 1 && a[0] == '-') return fail_usage("unexpected argument '" + a + "' found");
        if (!scene.empty()) return fail_usage("unexpected argument '" + a + "' found");
        scene = a;
    }
    if (scene.empty()) return fail_usage("the following required arguments were not provided:\n  <SCENE>");

    auto fail = [&](const char* what) {  // main() returning Err: message on stderr, exit status 1
        std::fprintf(stderr, "Error: %s: %s\n", what, rtm_last_error());
        return 1;
    };
    rtm_frontend* fe = nullptr;
    if (rtm_frontend_load(scene.c_str(), nullptr, base_dir.c_str(), synth ? 1u : 0u, &fe) != RTM_OK) return fail("scene");
    uint32_t width = 0, height = 0;
    rtm_frontend_size(fe, &width, &height);
    int status = 0;
    rtm_scene_dev* sd = nullptr;
    if (rtm_init(0) != RTM_OK) status = fail("device");
    if (!status && rtm_scene_upload(rtm_frontend_scene(fe), &sd) != RTM_OK) status = fail("scene upload");
    if (!status) {
        rtm_params p;
        std::memset(&p, 0, sizeof p);
        p.width = width; p.height = height; p.spp = (uint32_t)spp; p.seed = seed; p.max_bounces = 10;  // K, main.rs:63
        p.row_begin = 0; p.row_stride = 1;
        std::vector<uint8_t> rgb8((size_t)width * height * 3u);
        double seconds = 0.0;
        if (rtm_render_rgb8(sd, &p, gamma ? 1 : 0, rgb8.data(), &seconds) != RTM_OK) status = fail("render");
        if (!status) {
            std::printf("Rendered %s in %s\n", output.c_str(), rust_duration(seconds).c_str());  // main.rs:188
            std::fflush(stdout);
            if (rtm_write_bmp(output.c_str(), width, height, rgb8.data()) != RTM_OK) status = fail("save");
        }
    }
    if (sd) rtm_scene_free(sd);
    rtm_frontend_free(fe);
    return status;
}

}  // extern "C"
