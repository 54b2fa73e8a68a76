// C wrappers around the C++ host classes so that scripts (tests, the benchmark) can drive them
// through ctypes. Not part of the gpurt device ABI: these only forward to the classes next to them.

#include <cstring>
#include <iostream>
#include <vector>

#include "RankGate.h"
#include "RaytracerPipeline.h"
#include "Scenes.h"

using namespace RT;
using namespace RT::GraphicsAPI;

extern "C" {

struct rth_settings
{
	uint32_t width, height, max_rays_per_pixel, max_ray_depth;
	float aa_spread, dof_spread;
	int32_t use_bvh;
	float fov, near_z, far_z;
	float cam_position[3], cam_focus[3], cam_up[3];
	int32_t rebuild_every_cycle, fused_build;
	uint32_t seed_raygen, seed_trace;
	int32_t dynamic_scene;
	int32_t reference_order_walk;
	int32_t drop_missed_rays;
	int32_t no_previews;
	int32_t refit;
	uint32_t device_rank, device_count;
};

void rth_camera_matrices(const float* position, const float* focus, const float* up, float fov, float aspect, float near_z, float far_z,
	float* inverse_view, float* inverse_projection)
{
	CameraInfo c{};
	c.Position = { position[0], position[1], position[2] };
	c.FocusPoint = { focus[0], focus[1], focus[2] };
	c.UpDirection = { up[0], up[1], up[2] };
	c.VerticalFOV = fov; c.NearZ = near_z; c.FarZ = far_z;
	ComputeCameraMatrices(c, aspect, inverse_view, inverse_projection);
}

float rth_parse_real(const char* text) { return ParseReal(text, text + std::strlen(text)); }

// ---- meshes ----
MeshInfo* rth_load_mesh_dedup(const char* path)
{
	MeshInfo m = LoadMeshFromFile(path, true);
	if (!m.Vertices) return nullptr;
	return new MeshInfo(m);
}
MeshInfo* rth_load_mesh_threads(const char* path, int dedup, uint32_t threads)
{
	MeshInfo m = LoadMeshFromFile(path, dedup != 0, threads);
	std::cout.flush();		//the loader's messages, before the caller goes on writing to the same descriptor
	if (!m.Vertices) return nullptr;
	return new MeshInfo(m);
}
MeshInfo* rth_load_mesh(const char* path)
{
	MeshInfo m = LoadMeshFromFile(path);
	if (!m.Vertices) return nullptr;
	return new MeshInfo(m);
}
MeshInfo* rth_synthetic_scene(uint32_t n, uint32_t seed)
{
	MeshInfo m = GenerateSyntheticScene(n, seed);
	if (!m.Vertices) return nullptr;
	return new MeshInfo(m);
}
//a MeshInfo from arrays (a soup: indices 0 .. vertex_count - 1; no textures beyond the built-in slot 0), e.g. a loader fixture;
//the centroid bounds are computed as the loader computes them
MeshInfo* rth_mesh_from_arrays(const void* vertices, uint64_t vertex_count, const void* materials, uint64_t material_count)
{
	if (!vertices || !materials || vertex_count < 9 || vertex_count % 3 || !material_count) return nullptr;
	MeshInfo m{};
	m.IndexCount = m.VertexCount = vertex_count;
	m.Indices = new Index[vertex_count];
	for (uint64_t i = 0; i < vertex_count; i++) m.Indices[i] = (Index)i;
	m.Vertices = new Vertex[vertex_count];
	std::memcpy(m.Vertices, vertices, vertex_count * sizeof(Vertex));
	m.MaterialCount = material_count;
	m.Materials = new PBRMaterial[material_count];
	std::memcpy(m.Materials, materials, material_count * sizeof(PBRMaterial));
	m.TextureNameCount = 1;
	m.TextureNames = new std::string[1];
	m.SceneAABB = CalculateCentroidBounds(m);
	return new MeshInfo(m);
}
uint64_t rth_mesh_index_count(const MeshInfo* m) { return m->IndexCount; }
uint64_t rth_mesh_vertex_count(const MeshInfo* m) { return m->VertexCount; }
uint64_t rth_mesh_material_count(const MeshInfo* m) { return m->MaterialCount; }
uint64_t rth_mesh_texture_count(const MeshInfo* m) { return m->TextureNameCount; }
const char* rth_mesh_texture_name(const MeshInfo* m, uint64_t i) { return m->TextureNames[i].c_str(); }
void rth_mesh_copy(const MeshInfo* m, uint32_t* indices, void* vertices, void* materials, float* scene_min, float* scene_max)
{
	if (indices) std::memcpy(indices, m->Indices, m->IndexCount * sizeof(Index));
	if (vertices) std::memcpy(vertices, m->Vertices, m->VertexCount * sizeof(Vertex));
	if (materials) std::memcpy(materials, m->Materials, m->MaterialCount * sizeof(PBRMaterial));
	if (scene_min) { scene_min[0] = m->SceneAABB.Min.x; scene_min[1] = m->SceneAABB.Min.y; scene_min[2] = m->SceneAABB.Min.z; }
	if (scene_max) { scene_max[0] = m->SceneAABB.Max.x; scene_max[1] = m->SceneAABB.Max.y; scene_max[2] = m->SceneAABB.Max.z; }
}
void rth_mesh_free(MeshInfo* m) { if (m) { ReleaseMeshInfo(*m); delete m; } }

// ---- textures and atlas ----
//decodes to RGBA16; returns a buffer to release with rth_free, or null
uint16_t* rth_load_texture(const char* path, uint32_t* width, uint32_t* height)
{
	TextureInfo t = LoadTextureFromFile(path);
	if (!t.Data) return nullptr;
	*width = t.Width; *height = t.Height;
	return (uint16_t*)t.Data;
}
void rth_free(void* p) { delete[] (uint8_t*)p; }

//packs the textures named by a mesh (paths relative to the working directory) into the atlas host image;
//returns the byte size, or 0 on failure. Pass null outputs to query sizes.
uint64_t rth_build_atlas(const MeshInfo* m, const char* directory_prefix, uint8_t* atlas_bytes, uint64_t atlas_capacity, void* ids, uint64_t id_capacity)
{
	TextureAtlas atlas;
	for (uint64_t i = 1; i < m->TextureNameCount; i++)
	{
		uint32_t id = 0;
		std::string path = std::string(directory_prefix ? directory_prefix : "") + m->TextureNames[i];
		if (!atlas.AddTexture(&id, LoadTextureFromFile(path))) return 0;
	}
	std::vector<uint8_t> bytes;
	std::vector<TextureID> tids;
	atlas.BuildHostImage(bytes, tids);
	if (atlas_bytes && atlas_capacity >= bytes.size()) std::memcpy(atlas_bytes, bytes.data(), bytes.size());
	if (ids && id_capacity >= tids.size()) std::memcpy(ids, tids.data(), tids.size() * sizeof(TextureID));
	return ((uint64_t)tids.size() << 40) | (uint64_t)bytes.size();
}

// ---- pipeline ----
static Settings ToSettings(const rth_settings* s)
{
	Settings r;
	r.WindowWidth = s->width; r.WindowHeight = s->height;
	r.MaxRaysPerPixel = s->max_rays_per_pixel; r.MaxRayDepth = s->max_ray_depth;
	r.AASampleSpread = s->aa_spread; r.DOFSampleSpread = s->dof_spread;
	r.UseBVH = s->use_bvh != 0;
	r.CameraFOV = s->fov; r.CameraNearZ = s->near_z; r.CameraFarZ = s->far_z;
	for (int k = 0; k < 3; k++) { r.CameraPosition[k] = s->cam_position[k]; r.CameraFocusPoint[k] = s->cam_focus[k]; r.CameraUpDirection[k] = s->cam_up[k]; }
	r.RebuildBVHEveryCycle = s->rebuild_every_cycle != 0;
	r.FusedBuild = s->fused_build != 0;
	r.SeedRayGen = s->seed_raygen; r.SeedTrace = s->seed_trace;
	r.DynamicScene = s->dynamic_scene != 0;
	r.ReferenceOrderWalk = s->reference_order_walk != 0;
	r.DropMissedRays = s->drop_missed_rays != 0;
	r.PreviewEveryBounce = s->no_previews == 0;
	r.RefitBVH = s->refit != 0;
	r.DeviceRank = s->device_rank;
	r.DeviceCount = s->device_count ? s->device_count : 1;
	return r;
}

//consumes the mesh handle (the pipeline takes ownership of its arrays)
RaytracerPipeline* rth_pipeline_create(rt_device* dev, MeshInfo* mesh, const rth_settings* s)
{
	Settings settings = ToSettings(s);
	RaytracerPipeline* p = new RaytracerPipeline();
	MeshInfo m = *mesh;
	std::string* names = m.TextureNames;
	delete mesh;
	bool ok = p->Initialize(dev, m, &settings);
	delete[] names;
	if (!ok) { delete p; return nullptr; }
	return p;
}
int rth_pipeline_render(RaytracerPipeline* p, uint32_t calls)
{
	for (uint32_t i = 0; i < calls; i++) if (!p->Render()) return 0;
	return 1;
}
int rth_pipeline_read_output(RaytracerPipeline* p, float* pixels) { return p->ReadOutput(pixels) ? 1 : 0; }
int rth_pipeline_read_result(RaytracerPipeline* p, float* pixels) { return p->ReadResult(pixels) ? 1 : 0; }
rt_buffer* rth_pipeline_bvh(RaytracerPipeline* p) { return p->GetBVH(); }
rt_buffer* rth_pipeline_morton(RaytracerPipeline* p) { return p->GetMortonCodes(); }
rt_buffer* rth_pipeline_slot(RaytracerPipeline* p, uint32_t slot) { return slot < 8 ? p->GetDescriptorHeap()->Slots[slot] : nullptr; }
int rth_pipeline_save_state(RaytracerPipeline* p, const char* path) { return p->SaveState(path) ? 1 : 0; }
int rth_pipeline_load_state(RaytracerPipeline* p, const char* path) { return p->LoadState(path) ? 1 : 0; }
uint32_t rth_pipeline_num_samples(RaytracerPipeline* p) { return p->GetNumSamples(); }
uint32_t rth_pipeline_local_rows(RaytracerPipeline* p) { return p->GetLocalRows(); }
int rth_pipeline_gather_output(RaytracerPipeline* p, rt_comm* comm, rt_buffer* image) { return p->GatherOutput(comm, image) ? 1 : 0; }
void rth_pipeline_destroy(RaytracerPipeline* p) { delete p; }

int rth_write_ppm(const char* path, const float* pixels, uint32_t w, uint32_t h) { return WriteImagePPM(path, pixels, w, h) ? 1 : 0; }
int rth_write_pfm(const char* path, const float* pixels, uint32_t w, uint32_t h) { return WriteImagePFM(path, pixels, w, h) ? 1 : 0; }
int rth_write_png(const char* path, const float* pixels, uint32_t w, uint32_t h) { return WriteImagePNG(path, pixels, w, h) ? 1 : 0; }
uint32_t rth_encode_srgb8(float linear) { return EncodeSRGB8(linear); }

// ---- the gate the rank threads of rt_headless --ranks meet at before a collective (RankGate.h) ----
RankGate* rth_gate_create(int world) { return new RankGate(world); }
int rth_gate_arrive(RankGate* g, int well) { return g->Arrive(well != 0) ? 1 : 0; }
void rth_gate_leave(RankGate* g) { g->Leave(); }
int rth_gate_failed(RankGate* g) { return g->Failed() ? 1 : 0; }
void rth_gate_destroy(RankGate* g) { delete g; }

}
