// Headless entry point: loads (or generates) a scene, runs RaytracerPipeline::Render for a number of
// sample cycles and writes the image, in place of the reference's window + render thread
// (src/Main.cpp:29-87 of the reference).
//
//   rt_headless [--scene file.obj | --synthetic N] [--width W] [--height H] [--rpp R] [--depth D]
//               [--cycles C] [--no-bvh] [--per-shader-build] [--rebuild] [--camera px py pz fx fy fz]
//               [--seed S] [--gpu G] [--out image.ppm|image.png] [--pfm image.pfm] [--dynamic] [--refit] [--reference-walk]
//               [--resume state.bin] [--save-state state.bin] [--dedup] [--loader-threads T] [--drop-missed-rays] [--no-previews]
//               [--ranks N]
//
// --ranks N renders on GPUs G .. G + N - 1, one host thread and one RaytracerPipeline per GPU (the reference drives its
// pipeline from one render thread, src/Main.cpp:18-25): the image rows are dealt to the devices in turn, every device
// loads the scene and builds its own hierarchy, and the finished rows are gathered to the first device over RCCL.
// --ranks 1 takes the same path on one GPU.

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>
#include <vector>

#include "RaytracerPipeline.h"
#include "Scenes.h"

using namespace RT;
using namespace RT::GraphicsAPI;

namespace
{
	struct Options
	{
		Settings rtSettings;
		std::string sOut = "image.ppm", sPFM, sResume, sSaveState;
		uint32_t iSynthetic = 0, iCycles = 1;
		int iGPU = 0;
		int iRanks = 0;		//0: the single-device path without a communicator
	};

	std::mutex s_stdPrintLock;
	void Say(const std::string& s) { std::lock_guard<std::mutex> stdGuard(s_stdPrintLock); std::cout << s << "\n"; }

	bool WriteImages(const Options& rtOptions, const float* pPixels)
	{
		const Settings& rtSettings = rtOptions.rtSettings;
		const std::string& sOut = rtOptions.sOut;
		const bool bPNG = sOut.size() > 4 && sOut.compare(sOut.size() - 4, 4, ".png") == 0;
		if (!sOut.empty() && !(bPNG ? WriteImagePNG(sOut, pPixels, rtSettings.WindowWidth, rtSettings.WindowHeight)
		                              : WriteImagePPM(sOut, pPixels, rtSettings.WindowWidth, rtSettings.WindowHeight))) { Say("Error while writing " + sOut); return false; }
		if (!rtOptions.sPFM.empty() && !WriteImagePFM(rtOptions.sPFM, pPixels, rtSettings.WindowWidth, rtSettings.WindowHeight)) { Say("Error while writing " + rtOptions.sPFM); return false; }
		return true;
	}

	//one device's share of the run; iRank < 0: the whole image on one device, no communicator
	int RunDevice(const Options& rtOptions, int iRank, int iWorld, const void* pCommID)
	{
		Settings rtSettings = rtOptions.rtSettings;
		const bool bComm = iRank >= 0;
		if (bComm) { rtSettings.DeviceRank = (uint32_t)iRank; rtSettings.DeviceCount = (uint32_t)iWorld; }
		const bool bFirst = !bComm || iRank == 0;

		rt_device* rtDevice = nullptr;
		if (rt_device_create(rtOptions.iGPU + (bComm ? iRank : 0), &rtDevice) != RT_OK) { Say("Error while creating the device"); return 1; }
		if (bFirst) Say("Device was initialized successfully");

		MeshInfo rtMesh{};
		if (rtOptions.iSynthetic) { rtMesh = GenerateSyntheticScene(rtOptions.iSynthetic); ApplySyntheticCamera(rtSettings); }
		else rtMesh = LoadMeshFromFile(rtSettings.SceneFilename, rtSettings.DeduplicateVertices, rtSettings.LoaderThreads);
		if (!rtMesh.Vertices) { Say("Error while loading the scene"); return 1; }
		const uint64_t iTriangles = rtMesh.IndexCount / 3;

		rt_comm* rtComm = nullptr;
		if (bComm && rt_comm_init(rtDevice, iRank, iWorld, pCommID, &rtComm) != RT_OK) { Say("Error while joining the device group (RCCL)"); return 1; }

		int iResult = 0;
		{
			RaytracerPipeline rtRenderer;
			if (!rtRenderer.Initialize(rtDevice, rtMesh, &rtSettings)) { Say("Error while initializing the raytracing pipeline"); return 1; }
			if (bFirst) Say("Raytracing pipeline was initialized successfully");
			if (!rtOptions.sResume.empty())
			{
				if (!rtRenderer.LoadState(rtOptions.sResume)) { Say("Error while reading " + rtOptions.sResume); return 1; }
				Say("Resumed at " + std::to_string(rtRenderer.GetNumSamples() - 1) + " accumulated sample cycles");
			}

			const auto stdStart = std::chrono::steady_clock::now();
			uint64_t iCalls = 0;
			for (uint32_t c = 0; c < rtOptions.iCycles; c++)
			{
				for (uint32_t d = 0; d < rtSettings.MaxRayDepth; d++, iCalls++)
				{
					if (!rtRenderer.Render()) { Say("Error while rendering"); return 1; }
				}
				const float fSeconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - stdStart).count();
				if (!bComm && fSeconds > rtSettings.MaxSeconds) break;		//(a group renders the same number of cycles on every device)
			}

			const size_t iPixels = (size_t)rtSettings.WindowWidth * rtSettings.WindowHeight;
			std::vector<float> stdPixels;
			if (bComm)
			{
				rt_buffer* rtImage = nullptr;
				if (bFirst && rt_buffer_alloc(rtDevice, iPixels * 16, &rtImage) != RT_OK) { Say("Error while allocating the image"); return 1; }
				if (!rtRenderer.GatherOutput(rtComm, rtImage)) { Say("Error while gathering the image rows"); return 1; }
				rt_device_sync(rtDevice);
				if (bFirst)
				{
					stdPixels.resize(iPixels * 4);
					if (rt_buffer_readback(rtImage, stdPixels.data(), iPixels * 16, 0) != RT_OK) { Say("Error while reading the image"); return 1; }
					rt_buffer_free(rtImage);
				}
			}
			else
			{
				rt_device_sync(rtDevice);
				stdPixels.resize(iPixels * 4);
				if (!rtRenderer.ReadOutput(stdPixels.data())) { Say("Error while reading the image"); return 1; }
			}
			const float fSeconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - stdStart).count();
			if (bFirst)
			{
				const double fRays = (double)iCalls * rtSettings.WindowWidth * rtSettings.WindowHeight * rtSettings.MaxRaysPerPixel;
				Say(std::to_string(iTriangles) + " triangles, " + std::to_string(iCalls) + " bounces in " + std::to_string(fSeconds) + " s on " +
					std::to_string(bComm ? iWorld : 1) + " GPU(s): " + std::to_string(fRays / fSeconds * 1e-6) + " Mrays/s");
				if (!rtOptions.sSaveState.empty() && !rtRenderer.SaveState(rtOptions.sSaveState)) { Say("Error while writing " + rtOptions.sSaveState); iResult = 1; }
				if (!WriteImages(rtOptions, stdPixels.data())) iResult = 1;
			}
		}
		if (rtComm) rt_comm_destroy(rtComm);
		rt_device_destroy(rtDevice);
		return iResult;
	}
}

int main(int argc, char** argv)
{
	Options rtOptions;
	Settings& rtSettings = rtOptions.rtSettings;
	rtSettings.SeedRayGen = 1;
	rtSettings.SeedTrace = 2;
	for (int i = 1; i < argc; i++)
	{
		auto Next = [&]() -> const char* { if (i + 1 >= argc) { std::cout << "missing value after " << argv[i] << "\n"; std::exit(2); } return argv[++i]; };
		if (!std::strcmp(argv[i], "--scene")) rtSettings.SceneFilename = Next();
		else if (!std::strcmp(argv[i], "--synthetic")) rtOptions.iSynthetic = (uint32_t)std::atoll(Next());
		else if (!std::strcmp(argv[i], "--width")) rtSettings.WindowWidth = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--height")) rtSettings.WindowHeight = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--rpp")) rtSettings.MaxRaysPerPixel = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--depth")) rtSettings.MaxRayDepth = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--cycles")) rtOptions.iCycles = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--no-bvh")) rtSettings.UseBVH = false;
		else if (!std::strcmp(argv[i], "--per-shader-build")) rtSettings.FusedBuild = false;
		else if (!std::strcmp(argv[i], "--rebuild")) rtSettings.RebuildBVHEveryCycle = true;
		else if (!std::strcmp(argv[i], "--seed")) { rtSettings.SeedRayGen = (uint32_t)std::atoll(Next()); rtSettings.SeedTrace = rtSettings.SeedRayGen + 1; }
		else if (!std::strcmp(argv[i], "--gpu")) rtOptions.iGPU = std::atoi(Next());
		else if (!std::strcmp(argv[i], "--out")) rtOptions.sOut = Next();
		else if (!std::strcmp(argv[i], "--pfm")) rtOptions.sPFM = Next();
		else if (!std::strcmp(argv[i], "--dynamic")) rtSettings.DynamicScene = true;
		else if (!std::strcmp(argv[i], "--reference-walk")) rtSettings.ReferenceOrderWalk = true;
		else if (!std::strcmp(argv[i], "--refit")) rtSettings.RefitBVH = true;
		else if (!std::strcmp(argv[i], "--dedup")) rtSettings.DeduplicateVertices = true;
		else if (!std::strcmp(argv[i], "--loader-threads")) rtSettings.LoaderThreads = (uint32_t)std::atoi(Next());
		else if (!std::strcmp(argv[i], "--drop-missed-rays")) rtSettings.DropMissedRays = true;
		else if (!std::strcmp(argv[i], "--no-previews")) rtSettings.PreviewEveryBounce = false;
		else if (!std::strcmp(argv[i], "--resume")) rtOptions.sResume = Next();
		else if (!std::strcmp(argv[i], "--save-state")) rtOptions.sSaveState = Next();
		else if (!std::strcmp(argv[i], "--ranks")) rtOptions.iRanks = std::atoi(Next());
		else if (!std::strcmp(argv[i], "--camera"))
		{
			for (int k = 0; k < 3; k++) rtSettings.CameraPosition[k] = (float)std::atof(Next());
			for (int k = 0; k < 3; k++) rtSettings.CameraFocusPoint[k] = (float)std::atof(Next());
		}
		else { std::cout << "unknown option " << argv[i] << "\n"; return 2; }
	}

	if (rtOptions.iRanks <= 0) return RunDevice(rtOptions, -1, 1, nullptr);

	if (!rtOptions.sResume.empty() || !rtOptions.sSaveState.empty()) { std::cout << "--resume / --save-state need the single-device path\n"; return 2; }
	if (rtSettings.SeedRayGen == 0 || rtSettings.SeedTrace == 0) { std::cout << "--ranks needs fixed seeds: every device draws the same host seeds\n"; return 2; }
	unsigned char pCommID[RT_COMM_ID_BYTES];
	if (rt_comm_unique_id(pCommID) != RT_OK) { std::cout << "Error while creating the device group (RCCL)\n"; return 1; }
	std::vector<std::thread> stdThreads;
	std::atomic<int> iFailed{ 0 };
	for (int r = 0; r < rtOptions.iRanks; r++)
		stdThreads.emplace_back([&, r]() { if (RunDevice(rtOptions, r, rtOptions.iRanks, pCommID) != 0) iFailed++; });
	for (auto& stdThread : stdThreads) stdThread.join();
	return iFailed ? 1 : 0;
}
