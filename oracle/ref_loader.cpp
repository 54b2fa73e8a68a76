// Reference-loader checker (TEST INFRASTRUCTURE, built only where /root/reference is present).
//
// Compiles the third-party headers the reference vendors -- tinyobjloader 2.0.0 and stb_image 2.27 --
// in place from the read-only reference checkout (-I/root/reference/include; nothing is copied) and
// dumps what they produce for an OBJ/MTL file or a PNG file with exactly the calls and options the
// reference's loader uses (src/RaytracerMesh.cpp:23-31, 183-198). scripts/make_golden.py turns the
// dumps into the fixtures under tests/golden/ that pin the from-scratch OBJ/MTL/PNG code of the host.
//
//   ref_loader obj  <file.obj> <out-prefix>   -> <prefix>.corners.bin  (float x8 + int32 per face corner:
//                                                 position, uv, normal, material id), <prefix>.materials.txt
//   ref_loader png  <file.png> <out.bin>      -> uint32 width, height, then RGBA16 pixels

#define TINYOBJLOADER_IMPLEMENTATION
#include <tinyobjloader/tiny_obj_loader.h>
#define STB_IMAGE_IMPLEMENTATION
#include <stb/stb_image.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

static int dump_obj(const char* path, const std::string& prefix) {
    tinyobj::ObjReaderConfig cfg{};
    cfg.triangulate = false;
    cfg.triangulation_method = "simple";
    cfg.vertex_color = false;
    cfg.mtl_search_path = "";
    tinyobj::ObjReader reader{};
    if (!reader.ParseFromFile(path, cfg)) { std::fprintf(stderr, "parse failed: %s\n", reader.Error().c_str()); return 1; }
    const auto& attrib = reader.GetAttrib();
    const auto& shapes = reader.GetShapes();
    const auto& materials = reader.GetMaterials();
    FILE* f = std::fopen((prefix + ".corners.bin").c_str(), "wb");
    if (!f) return 1;
    for (const auto& shape : shapes) {
        for (size_t face = 0; face < shape.mesh.num_face_vertices.size(); face++) {
            if (shape.mesh.num_face_vertices[face] != 3) { std::fprintf(stderr, "non-triangle face\n"); return 1; }
            for (int v = 0; v < 3; v++) {
                const auto& idx = shape.mesh.indices[3 * face + v];
                float rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int k = 0; k < 3; k++) rec[k] = attrib.vertices[3 * idx.vertex_index + k];
                if (idx.texcoord_index >= 0) for (int k = 0; k < 2; k++) rec[3 + k] = attrib.texcoords[2 * idx.texcoord_index + k];
                if (idx.normal_index >= 0) for (int k = 0; k < 3; k++) rec[5 + k] = attrib.normals[3 * idx.normal_index + k];
                int32_t mat = shape.mesh.material_ids[face];
                std::fwrite(rec, sizeof(float), 8, f);
                std::fwrite(&mat, sizeof(int32_t), 1, f);
            }
        }
    }
    std::fclose(f);
    f = std::fopen((prefix + ".materials.txt").c_str(), "w");
    if (!f) return 1;
    for (const auto& m : materials) {
        // hex floats keep every bit
        std::fprintf(f, "%a %a %a  %a %a %a  %a %a %a  %a %a %a\n", m.diffuse[0], m.diffuse[1], m.diffuse[2], m.specular[0], m.specular[1],
                     m.specular[2], m.emission[0], m.emission[1], m.emission[2], m.shininess, m.roughness, m.metallic);
        std::fprintf(f, "|%s|%s|%s|%s|%s|\n", m.diffuse_texname.c_str(), m.roughness_texname.c_str(), m.specular_texname.c_str(),
                     m.metallic_texname.c_str(), m.emissive_texname.c_str());
    }
    std::fclose(f);
    return 0;
}

static int dump_png(const char* path, const char* out) {
    int w = 0, h = 0, n = 0;
    stbi_us* px = stbi_load_16(path, &w, &h, &n, 4);
    if (!px) { std::fprintf(stderr, "decode failed: %s\n", stbi_failure_reason()); return 1; }
    FILE* f = std::fopen(out, "wb");
    if (!f) return 1;
    uint32_t dims[2] = {(uint32_t)w, (uint32_t)h};
    std::fwrite(dims, 4, 2, f);
    std::fwrite(px, 2, (size_t)w * h * 4, f);
    std::fclose(f);
    stbi_image_free(px);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 4 && !std::strcmp(argv[1], "obj")) return dump_obj(argv[2], argv[3]);
    if (argc == 4 && !std::strcmp(argv[1], "png")) return dump_png(argv[2], argv[3]);
    std::fprintf(stderr, "usage: ref_loader obj <file.obj> <out-prefix> | ref_loader png <file.png> <out.bin>\n");
    return 2;
}
