#include "Scenes.h"

#include <cmath>
#include <cstring>
#include <random>

namespace RT::GraphicsAPI
{
	namespace
	{
		inline Float3 Sub(Float3 a, Float3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
		inline Float3 Cross(Float3 a, Float3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
		inline Float3 UnitOrSelf(Float3 v)
		{
			float n = std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
			if (n == 0.0f) n = 1.0f;
			return { v.x / n, v.y / n, v.z / n };
		}
	}

	MeshInfo GenerateSyntheticScene(uint32_t iNumTriangles, uint32_t iSeed)
	{
		MeshInfo rtMesh{};
		if (iNumTriangles < 15) return rtMesh;
		const uint32_t n = iNumTriangles, iRandom = n - 12;
		rtMesh.IndexCount = rtMesh.VertexCount = 3ull * n;
		rtMesh.Indices = new Index[rtMesh.IndexCount];
		rtMesh.Vertices = new Vertex[rtMesh.VertexCount];
		std::memset(rtMesh.Vertices, 0, sizeof(Vertex) * rtMesh.VertexCount);

		std::mt19937 stdGenerator(iSeed);
		auto Unit = [&]() { return (float)((uint32_t)stdGenerator() >> 8) * 5.9604644775390625e-08f; };
		const float e = 20.0f / (float)std::cbrt((double)n);
		for (uint32_t t = 0; t < iRandom; t++)
		{
			float r[9];
			for (int k = 0; k < 9; k++) r[k] = Unit();
			const Float3 c = { -10.0f + 20.0f * r[0], -10.0f + 20.0f * r[1], -10.0f + 20.0f * r[2] };
			const Float3 e1 = { e * (2.0f * r[3] - 1.0f), e * (2.0f * r[4] - 1.0f), e * (2.0f * r[5] - 1.0f) };
			const Float3 e2 = { e * (2.0f * r[6] - 1.0f), e * (2.0f * r[7] - 1.0f), e * (2.0f * r[8] - 1.0f) };
			rtMesh.Vertices[3 * t].Position = c;
			rtMesh.Vertices[3 * t + 1].Position = { c.x + e1.x, c.y + e1.y, c.z + e1.z };
			rtMesh.Vertices[3 * t + 2].Position = { c.x + e2.x, c.y + e2.y, c.z + e2.z };
		}
		//the enclosing box, two triangles per face
		const float h = 25.0f;
		const Float3 pCorners[8] = { { -h, -h, -h }, { h, -h, -h }, { h, h, -h }, { -h, h, -h }, { -h, -h, h }, { h, -h, h }, { h, h, h }, { -h, h, h } };
		const int pQuads[6][4] = { { 0, 1, 2, 3 }, { 5, 4, 7, 6 }, { 4, 0, 3, 7 }, { 1, 5, 6, 2 }, { 3, 2, 6, 7 }, { 4, 5, 1, 0 } };
		for (int q = 0; q < 6; q++)
		{
			const uint32_t t = iRandom + 2 * q;
			rtMesh.Vertices[3 * t].Position = pCorners[pQuads[q][0]];
			rtMesh.Vertices[3 * t + 1].Position = pCorners[pQuads[q][1]];
			rtMesh.Vertices[3 * t + 2].Position = pCorners[pQuads[q][2]];
			rtMesh.Vertices[3 * t + 3].Position = pCorners[pQuads[q][0]];
			rtMesh.Vertices[3 * t + 4].Position = pCorners[pQuads[q][2]];
			rtMesh.Vertices[3 * t + 5].Position = pCorners[pQuads[q][3]];
		}
		const Float2 pUV[3] = { { 0.0f, 0.0f }, { 1.0f, 0.0f }, { 0.0f, 1.0f } };
		for (uint32_t t = 0; t < n; t++)
		{
			Vertex* v = &rtMesh.Vertices[3 * t];
			const Float3 d1 = Sub(v[1].Position, v[0].Position), d2 = Sub(v[2].Position, v[0].Position);
			const Float3 xmNormal = UnitOrSelf(Cross(d1, d2)), xmTangent = UnitOrSelf(d1);
			for (int k = 0; k < 3; k++)
			{
				v[k].UV = pUV[k];
				v[k].Normal = xmNormal;
				v[k].Tangent = xmTangent;
				v[k].MaterialID = t < iRandom ? (t % 8u) : 0u;
				rtMesh.Indices[3 * t + k] = 3 * t + k;
			}
		}

		rtMesh.MaterialCount = 8;
		rtMesh.Materials = new PBRMaterial[8];
		std::memset(rtMesh.Materials, 0, sizeof(PBRMaterial) * 8);
		const float pAlbedo[8][3] = { { 0.8f, 0.8f, 0.8f }, { 0.8f, 0.2f, 0.2f }, { 0.2f, 0.8f, 0.2f }, { 0.2f, 0.2f, 0.8f }, { 0.8f, 0.8f, 0.2f },
			{ 0.2f, 0.8f, 0.8f }, { 0.8f, 0.2f, 0.8f }, { 0.8f, 0.8f, 0.8f } };
		for (int i = 0; i < 8; i++)
		{
			rtMesh.Materials[i].Albedo = { pAlbedo[i][0], pAlbedo[i][1], pAlbedo[i][2] };
			rtMesh.Materials[i].Roughness = 0.5f;
			rtMesh.Materials[i].F0Color = { 0.04f, 0.04f, 0.04f };
		}
		rtMesh.Materials[7].Emissive = { 10.0f, 10.0f, 10.0f };
		rtMesh.TextureNameCount = 1;
		rtMesh.TextureNames = new std::string[1];
		rtMesh.SceneAABB = CalculateCentroidBounds(rtMesh);
		return rtMesh;
	}

	void ApplySyntheticCamera(Settings& rtSettings)
	{
		rtSettings.CameraPosition[0] = 0.0f; rtSettings.CameraPosition[1] = 0.0f; rtSettings.CameraPosition[2] = -22.0f;
		rtSettings.CameraFocusPoint[0] = rtSettings.CameraFocusPoint[1] = rtSettings.CameraFocusPoint[2] = 0.0f;
		rtSettings.CameraUpDirection[0] = 0.0f; rtSettings.CameraUpDirection[1] = 1.0f; rtSettings.CameraUpDirection[2] = 0.0f;
		rtSettings.CameraFOV = 1.2f; rtSettings.CameraNearZ = 0.01f; rtSettings.CameraFarZ = 10000.0f;
	}
}
