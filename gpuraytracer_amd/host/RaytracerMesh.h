#pragma once
// Scene model of the host surface: the reference's MeshInfo / Vertex / PBRMaterial / AABB / TextureInfo
// (src/RaytracerMesh.h:14-72 of the reference) with plain float arrays in place of the Windows math
// types, and a RaytracerMesh that owns the index and vertex device buffers through the gpurt C-ABI.

#include <cstdint>
#include <string>

#include "gpurt.h"

namespace RT::GraphicsAPI
{
	struct Float2 { float x, y; };
	struct Float3 { float x, y, z; };

	struct AABB
	{
		Float3 Min;
		Float3 Max;
		uint32_t Padding[2];
	};

	typedef uint32_t Index;

	struct Vertex
	{
		Float3 Position;
		Float2 UV;
		Float3 Normal;
		Float3 Tangent;
		uint32_t MaterialID;
	};

	struct PBRMaterial
	{
		Float3 Albedo;
		float Roughness;
		Float3 F0Color;
		float Metallic;
		Float3 Emissive;

		uint32_t AlbedoTextureID;
		uint32_t RoughnessTextureID;
		uint32_t F0TextureID;
		uint32_t MetallicTextureID;
		uint32_t EmissiveTextureID;
	};

	struct TextureInfo
	{
		void* Data;
		uint32_t Width;
		uint32_t Height;
		uint16_t BytesPerChannel;
		uint16_t ChannelCount;
	};

	struct MeshInfo
	{
		uint64_t IndexCount;
		Index* Indices;
		uint64_t VertexCount;
		Vertex* Vertices;
		uint64_t MaterialCount;
		PBRMaterial* Materials;
		uint64_t TextureNameCount;
		std::string* TextureNames;
		AABB SceneAABB;
	};

	static_assert(sizeof(Vertex) == sizeof(rt_vertex) && sizeof(PBRMaterial) == sizeof(rt_pbr_material) && sizeof(AABB) == sizeof(rt_aabb),
		"host structs must keep the device layouts");


	//decodes a PNG to tightly packed pixels; bHighPrecision selects 16 bits per channel (8-bit files are widened by
	//replicating the byte). A failed load returns a TextureInfo of zeros.
	TextureInfo LoadTextureFromFile(const std::string& sFileName, int iDesiredNumChannels = 4, bool bHighPrecision = true);
	//Wavefront OBJ + MTL. A failed load returns a MeshInfo of zeros. Only triangle faces are accepted.
	//bDeduplicate: equal vertices (position, UV, normal and material, compared as the reference's VerticesAreEqual does,
	//src/RaytracerMesh.cpp:157-169 of the reference) are stored once and shared through the index buffer - the loop the
	//reference has commented out as too slow (src/RaytracerMesh.cpp:249-275), done with a hash table. Default: off, as there.
	//iThreads: 1 = the file is parsed line by line as the reference's parser does; more = by that many threads (same
	//MeshInfo, bit for bit); 0 = chosen by the size of the file and the cores there are
	MeshInfo LoadMeshFromFile(const std::string& sFileName, bool bDeduplicate = false, uint32_t iThreads = 0);
	//frees the host arrays of a MeshInfo that was not handed to RaytracerMesh::Initialize
	void ReleaseMeshInfo(MeshInfo& rtMesh);

	//helpers of the loader, exposed for tests
	void CalculateNormals(MeshInfo* ionMeshInfo);
	void CalculateTangents(MeshInfo* ionMeshInfo);
	AABB CalculateCentroidBounds(const MeshInfo& rtMesh);
	float ParseReal(const char* pBegin, const char* pEnd, double fDefault = 0.0);


	class RaytracerMesh
	{
	private:

		rt_device* m_rtDevice;
		rt_buffer* m_rtIndexBuffer;
		rt_buffer* m_rtVertexBuffer;
		MeshInfo m_rtMesh;

	public:

		RaytracerMesh();
		~RaytracerMesh();

		//uploads indices and vertices; takes ownership of and deletes rtMesh.Indices / Vertices / Materials
		//afterwards, as the reference does
		//(bTakeOwnership = false: the arrays stay the caller's, e.g. for the next device's pipeline)
		bool Initialize(rt_device* rtDevice, MeshInfo rtMesh, bool bTakeOwnership = true);
		//the two resources a shader binds at t0 / t1
		void Bind(const rt_buffer** rtIndexBuffer, const rt_buffer** rtVertexBuffer) const;
		void Release();

		uint64_t GetIndexCount() { return m_rtMesh.IndexCount; };
		uint64_t GetVertexCount() { return m_rtMesh.VertexCount; };
	};
}
