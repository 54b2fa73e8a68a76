#pragma once
// Deterministic synthetic scene for the large benchmark configurations (the reference ships none):
// N - 12 small random triangles in [-10, 10]^3, eight materials round-robin (material 7 is a light), and
// a 12-triangle box [-25, 25]^3 around triangles and camera. Values come from std::mt19937(seed)
// mapped to [0, 1) by (u >> 8) * 2^-24, all arithmetic in binary32.

#include "RaytracerMesh.h"
#include "Settings.h"

namespace RT::GraphicsAPI
{
	//returns a MeshInfo of zeros if iNumTriangles < 15
	MeshInfo GenerateSyntheticScene(uint32_t iNumTriangles, uint32_t iSeed = 12345);
	//camera used with the synthetic scene: (0, 0, -22) looking at the origin
	void ApplySyntheticCamera(Settings& rtSettings);
}
