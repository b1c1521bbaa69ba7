// ref_inputs_dump.cpp -- TEST INFRASTRUCTURE.  Pins the INPUT path of the sample scene with the reference's own code.
//
// The reference's render path cannot be built here (Win32 + D3D12 + DXR + DXC: DESIGN.md 2), but the two single-header libraries its sample
// application loads its assets with are portable C / C++:
//     /root/reference/src/sample/contrib/stb_image.h        stbi_load(path, &w, &h, &n, STBI_rgb_alpha)      (src/sample/main.cpp:155-171)
//     /root/reference/src/sample/contrib/tiny_obj_loader.h  tinyobj::LoadObj(..., "res/sphere.obj", NULL, true) (src/sample/main.cpp:262-289)
// This file includes them UNMODIFIED from where they lie (oracle/ref_inputs.mk passes -I/root/reference/src/sample/contrib; nothing of the
// reference is copied into the repository) and writes what they return for the sample's assets into oracle/_ref/ (git-ignored):
//     <texture>.rgba   the w * h * 4 bytes stbi_load returns (only with --raw; index.json always carries their size and FNV-1a 64)
//     sphere.posnrm    per unrolled face vertex, in shape / face / corner order: position.xyz, normal.xyz as float32 (what the loop of main.cpp:269-287
//                      reads out of attrib.vertices / attrib.normals through shapes[i].mesh.indices)
//     index.json       sizes + FNV-1a 64 of every file (tests/golden/ref_inputs.json is a copy: the GPU box has no /root/reference)
// tests/test_ref_inputs.py compares the Python harness's loaders (sample_scene._load_png_rgba8, load_obj_unrolled) and the C host's own PNG / OBJ readers
// (tools/sample_host.c) with these bytes.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

static uint64_t fnv1a(const void *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
static bool write_file(const std::string &path, const void *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(p, 1, n, f) == n;
    fclose(f);
    return ok;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <res dir> <out dir> [--raw] <file.png | file.obj> ...\n", argv[0]); return 2; }
    const std::string res = argv[1], out = argv[2];
    std::string json = "{";
    bool raw = false, first = true;       // --raw: also keep the decoded texels (<name>.rgba, 45 MB for the sample's textures) -- the test asks for them only to locate a mismatch
    for (int a = 3; a < argc; a++) {
        const std::string name = argv[a], path = res + "/" + name;
        if (name == "--raw") { raw = true; continue; }
        if (!first) json += ", ";
        first = false;
        if (name.size() > 4 && name.substr(name.size() - 4) == ".obj") {
            tinyobj::attrib_t attrib; std::vector<tinyobj::shape_t> shapes; std::vector<tinyobj::material_t> materials; std::string warn, err;
            if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, path.c_str(), NULL, true)) { fprintf(stderr, "LoadObj(%s): %s\n", path.c_str(), err.c_str()); return 3; }
            std::vector<float> pn;
            for (size_t i = 0; i < shapes.size(); i++) {
                size_t at = 0;
                for (size_t f = 0; f < shapes[i].mesh.num_face_vertices.size(); f++) {
                    const size_t corners = shapes[i].mesh.num_face_vertices[f];
                    for (size_t v = 0; v < corners; v++) {
                        const tinyobj::index_t idx = shapes[i].mesh.indices[at + v];
                        for (int k = 0; k < 3; k++) pn.push_back(attrib.vertices[3 * idx.vertex_index + k]);
                        for (int k = 0; k < 3; k++) pn.push_back(attrib.normals[3 * idx.normal_index + k]);
                    }
                    at += corners;
                }
            }
            if (!write_file(out + "/sphere.posnrm", pn.data(), pn.size() * sizeof(float))) return 4;
            char buf[256]; snprintf(buf, sizeof(buf), "\"%s\": {\"vertices\": %zu, \"fnv1a\": \"%016llx\"}", name.c_str(), pn.size() / 6, (unsigned long long)fnv1a(pn.data(), pn.size() * sizeof(float)));
            json += buf;
        }
        else {
            int w = 0, h = 0, n = 0;
            unsigned char *px = stbi_load(path.c_str(), &w, &h, &n, STBI_rgb_alpha);
            if (!px) { fprintf(stderr, "stbi_load(%s): %s\n", path.c_str(), stbi_failure_reason()); return 3; }
            const size_t bytes = (size_t)w * h * 4;
            if (raw && !write_file(out + "/" + name + ".rgba", px, bytes)) return 4;
            char buf[256]; snprintf(buf, sizeof(buf), "\"%s\": {\"width\": %d, \"height\": %d, \"channels_in_file\": %d, \"fnv1a\": \"%016llx\"}", name.c_str(), w, h, n, (unsigned long long)fnv1a(px, bytes));
            json += buf;
            stbi_image_free(px);
        }
    }
    json += "}\n";
    if (!write_file(out + "/index.json", json.data(), json.size())) return 4;
    fputs(json.c_str(), stdout);
    return 0;
}
