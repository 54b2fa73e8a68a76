// Headless entry point: loads (or generates) a scene, runs RaytracerPipeline::Render for a number of
// sample cycles and writes the image, in place of the reference's window + render thread
// (src/Main.cpp:29-87 of the reference).
//
//   rt_headless [--scene file.obj | --synthetic N] [--width W] [--height H] [--rpp R] [--depth D]
//               [--cycles C] [--no-bvh] [--per-shader-build] [--rebuild] [--camera px py pz fx fy fz]
//               [--seed S] [--gpu G] [--out image.ppm|image.png] [--pfm image.pfm] [--dynamic] [--reference-walk]
//               [--resume state.bin] [--save-state state.bin]

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "RaytracerPipeline.h"
#include "Scenes.h"

using namespace RT;
using namespace RT::GraphicsAPI;

int main(int argc, char** argv)
{
	Settings rtSettings;
	std::string sOut = "image.ppm", sPFM, sResume, sSaveState;
	uint32_t iSynthetic = 0, iCycles = 1;
	int iGPU = 0;
	rtSettings.SeedRayGen = 1;
	rtSettings.SeedTrace = 2;
	for (int i = 1; i < argc; i++)
	{
		auto Next = [&]() -> const char* { if (i + 1 >= argc) { std::cout << "missing value after " << argv[i] << "\n"; std::exit(2); } return argv[++i]; };
		if (!std::strcmp(argv[i], "--scene")) rtSettings.SceneFilename = Next();
		else if (!std::strcmp(argv[i], "--synthetic")) iSynthetic = (uint32_t)std::atoll(Next());
		else if (!std::strcmp(argv[i], "--width")) rtSettings.WindowWidth = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--height")) rtSettings.WindowHeight = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--rpp")) rtSettings.MaxRaysPerPixel = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--depth")) rtSettings.MaxRayDepth = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--cycles")) iCycles = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--no-bvh")) rtSettings.UseBVH = false;
		else if (!std::strcmp(argv[i], "--per-shader-build")) rtSettings.FusedBuild = false;
		else if (!std::strcmp(argv[i], "--rebuild")) rtSettings.RebuildBVHEveryCycle = true;
		else if (!std::strcmp(argv[i], "--seed")) { rtSettings.SeedRayGen = (uint32_t)std::atoll(Next()); rtSettings.SeedTrace = rtSettings.SeedRayGen + 1; }
		else if (!std::strcmp(argv[i], "--gpu")) iGPU = std::atoi(Next());
		else if (!std::strcmp(argv[i], "--out")) sOut = Next();
		else if (!std::strcmp(argv[i], "--pfm")) sPFM = Next();
		else if (!std::strcmp(argv[i], "--dynamic")) rtSettings.DynamicScene = true;
		else if (!std::strcmp(argv[i], "--reference-walk")) rtSettings.ReferenceOrderWalk = true;
		else if (!std::strcmp(argv[i], "--resume")) sResume = Next();
		else if (!std::strcmp(argv[i], "--save-state")) sSaveState = Next();
		else if (!std::strcmp(argv[i], "--camera"))
		{
			for (int k = 0; k < 3; k++) rtSettings.CameraPosition[k] = (float)std::atof(Next());
			for (int k = 0; k < 3; k++) rtSettings.CameraFocusPoint[k] = (float)std::atof(Next());
		}
		else { std::cout << "unknown option " << argv[i] << "\n"; return 2; }
	}

	rt_device* rtDevice = nullptr;
	if (rt_device_create(iGPU, &rtDevice) != RT_OK)
	{
		std::cout << "Error while creating the device\n";
		return 1;
	}
	std::cout << "Device was initialized successfully\n";

	MeshInfo rtMesh{};
	if (iSynthetic) { rtMesh = GenerateSyntheticScene(iSynthetic); ApplySyntheticCamera(rtSettings); }
	else rtMesh = LoadMeshFromFile(rtSettings.SceneFilename);
	if (!rtMesh.Vertices)
	{
		std::cout << "Error while loading the scene\n";
		return 1;
	}
	const uint64_t iTriangles = rtMesh.IndexCount / 3;

	RaytracerPipeline rtRenderer;
	if (!rtRenderer.Initialize(rtDevice, rtMesh, &rtSettings))
	{
		std::cout << "Error while initializing the raytracing pipeline\n";
		return 1;
	}
	std::cout << "Raytracing pipeline was initialized successfully\n";
	if (!sResume.empty())
	{
		if (!rtRenderer.LoadState(sResume)) { std::cout << "Error while reading " << sResume << "\n"; return 1; }
		std::cout << "Resumed at " << rtRenderer.GetNumSamples() - 1 << " accumulated sample cycles\n";
	}

	const auto stdStart = std::chrono::steady_clock::now();
	uint64_t iCalls = 0;
	for (uint32_t c = 0; c < iCycles; c++)
	{
		for (uint32_t d = 0; d < rtSettings.MaxRayDepth; d++, iCalls++)
		{
			if (!rtRenderer.Render()) { std::cout << "Error while rendering\n"; return 1; }
		}
		const float fSeconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - stdStart).count();
		if (fSeconds > rtSettings.MaxSeconds) break;
	}
	rt_device_sync(rtDevice);
	const float fSeconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - stdStart).count();
	const double fRays = (double)iCalls * rtSettings.WindowWidth * rtSettings.WindowHeight * rtSettings.MaxRaysPerPixel;
	std::cout << iTriangles << " triangles, " << iCalls << " bounces in " << fSeconds << " s: " << fRays / fSeconds * 1e-6 << " Mrays/s\n";

	if (!sSaveState.empty() && !rtRenderer.SaveState(sSaveState)) { std::cout << "Error while writing " << sSaveState << "\n"; return 1; }
	std::vector<float> stdPixels((size_t)rtSettings.WindowWidth * rtSettings.WindowHeight * 4);
	if (!rtRenderer.ReadOutput(stdPixels.data())) { std::cout << "Error while reading the image\n"; return 1; }
	const bool bPNG = sOut.size() > 4 && sOut.compare(sOut.size() - 4, 4, ".png") == 0;
	if (!sOut.empty() && !(bPNG ? WriteImagePNG(sOut, stdPixels.data(), rtSettings.WindowWidth, rtSettings.WindowHeight)
	                              : WriteImagePPM(sOut, stdPixels.data(), rtSettings.WindowWidth, rtSettings.WindowHeight))) { std::cout << "Error while writing " << sOut << "\n"; return 1; }
	if (!sPFM.empty() && !WriteImagePFM(sPFM, stdPixels.data(), rtSettings.WindowWidth, rtSettings.WindowHeight)) { std::cout << "Error while writing " << sPFM << "\n"; return 1; }
	return 0;
}
