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
// pipeline from one render thread, src/Main.cpp:18-25): the image rows are dealt to the devices in turn, the scene is loaded
// once and uploaded to every device, which builds its own hierarchy, and the finished rows are gathered to the first device
// over RCCL. The ranks meet at a gate before every collective call: a rank that has failed keeps the others out of it and the
// process ends with a non-zero code instead of waiting for ever. --ranks 1 takes the same path on one GPU.

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>
#include <vector>

#include "RankGate.h"
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

	//A collective call that fails on one rank leaves the others inside it: nothing to return to. The process ends at once, with
	//the error code a caller can see.
	[[noreturn]] void GiveUp(const std::string& s)
	{
		Say(s + " - ending the process");
		std::cout.flush();
		std::_Exit(1);
	}

	//one device's share of the run; iRank < 0: the whole image on one device, no communicator.
	//rtGate (groups only): where the ranks meet before every collective, so that a rank that has failed keeps the others out of it
	int RunDevice(const Options& rtOptions, int iRank, int iWorld, const void* pCommID, const MeshInfo& rtMesh, RankGate* rtGate)
	{
		Settings rtSettings = rtOptions.rtSettings;
		const bool bComm = iRank >= 0;
		if (bComm) { rtSettings.DeviceRank = (uint32_t)iRank; rtSettings.DeviceCount = (uint32_t)iWorld; rtSettings.SharedMesh = true; }
		const bool bFirst = !bComm || iRank == 0;
		const uint64_t iTriangles = rtMesh.IndexCount / 3;

		rt_device* rtDevice = nullptr;
		rt_comm* rtComm = nullptr;
		//every way out that is not the last line: say why, tell the group, free what this rank holds
		auto Fail = [&](const std::string& s) -> int
		{
			Say(s + (bComm ? " (rank " + std::to_string(iRank) + ")" : ""));
			if (rtGate) rtGate->Leave();
			if (rtComm) rt_comm_destroy(rtComm);
			if (rtDevice) rt_device_destroy(rtDevice);
			return 1;
		};

		if (rt_device_create(rtOptions.iGPU + (bComm ? iRank : 0), &rtDevice) != RT_OK) return Fail("Error while creating the device");
		if (bFirst) Say("Device was initialized successfully");

		if (bComm)
		{
			if (!rtGate->Arrive()) return Fail("Another device of the group has failed");
			if (rt_comm_init(rtDevice, iRank, iWorld, pCommID, &rtComm) != RT_OK) GiveUp("Error while joining the device group (RCCL)");
		}

		int iResult = 0;
		{
			RaytracerPipeline rtRenderer;
			if (!rtRenderer.Initialize(rtDevice, rtMesh, &rtSettings)) return Fail("Error while initializing the raytracing pipeline");
			if (bFirst) Say("Raytracing pipeline was initialized successfully");
			if (!rtOptions.sResume.empty())
			{
				if (!rtRenderer.LoadState(rtOptions.sResume)) return Fail("Error while reading " + rtOptions.sResume);
				Say("Resumed at " + std::to_string(rtRenderer.GetNumSamples() - 1) + " accumulated sample cycles");
			}

			const auto stdStart = std::chrono::steady_clock::now();
			uint64_t iCalls = 0;
			bool bRendered = true;
			for (uint32_t c = 0; c < rtOptions.iCycles && bRendered; c++)
			{
				for (uint32_t d = 0; d < rtSettings.MaxRayDepth && bRendered; d++, iCalls++)
					bRendered = rtRenderer.Render();
				const float fSeconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - stdStart).count();
				if (!bComm && fSeconds > rtSettings.MaxSeconds) break;		//(a group renders the same number of cycles on every device)
				if (bComm && rtGate->Failed()) break;						//(nobody will gather this image)
			}
			if (!bRendered && !bComm) return Fail("Error while rendering");

			const size_t iPixels = (size_t)rtSettings.WindowWidth * rtSettings.WindowHeight;
			std::vector<float> stdPixels;
			if (bComm)
			{
				rt_buffer* rtImage = nullptr;
				bool bWell = bRendered;
				if (bWell && bFirst && rt_buffer_alloc(rtDevice, iPixels * 16, &rtImage) != RT_OK) bWell = false;
				if (!rtGate->Arrive(bWell))
				{
					if (rtImage) rt_buffer_free(rtImage);
					return Fail(bWell ? "Another device of the group has failed" : "Error while rendering");
				}
				if (!rtRenderer.GatherOutput(rtComm, rtImage)) GiveUp("Error while gathering the image rows");
				rt_device_sync(rtDevice);
				if (bFirst)
				{
					stdPixels.resize(iPixels * 4);
					const bool bRead = rt_buffer_readback(rtImage, stdPixels.data(), iPixels * 16, 0) == RT_OK;
					rt_buffer_free(rtImage);
					if (!bRead) return Fail("Error while reading the image");
				}
			}
			else
			{
				rt_device_sync(rtDevice);
				stdPixels.resize(iPixels * 4);
				if (!rtRenderer.ReadOutput(stdPixels.data())) return Fail("Error while reading the image");
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

	if (rtOptions.iRanks > 0)
	{
		if (!rtOptions.sResume.empty() || !rtOptions.sSaveState.empty()) { std::cout << "--resume / --save-state need the single-device path\n"; return 2; }
		if (rtSettings.SeedRayGen == 0 || rtSettings.SeedTrace == 0) { std::cout << "--ranks needs fixed seeds: every device draws the same host seeds\n"; return 2; }
		int iDevices = 0;
		if (rt_device_count(&iDevices) != RT_OK) { std::cout << "Error while counting the devices\n"; return 1; }
		if (rtOptions.iGPU < 0 || rtOptions.iGPU + rtOptions.iRanks > iDevices)
		{
			std::cout << "--gpu " << rtOptions.iGPU << " --ranks " << rtOptions.iRanks << " needs GPUs " << rtOptions.iGPU << " .. " << rtOptions.iGPU + rtOptions.iRanks - 1
			          << "; this machine shows " << iDevices << "\n";
			return 2;
		}
	}

	//the scene is loaded (or generated) once; the pipelines of a device group all upload from the same host arrays
	MeshInfo rtMesh{};
	if (rtOptions.iSynthetic) { rtMesh = GenerateSyntheticScene(rtOptions.iSynthetic); ApplySyntheticCamera(rtSettings); }
	else rtMesh = LoadMeshFromFile(rtSettings.SceneFilename, rtSettings.DeduplicateVertices, rtSettings.LoaderThreads);
	if (!rtMesh.Vertices) { std::cout << "Error while loading the scene\n"; return 1; }

	if (rtOptions.iRanks <= 0) return RunDevice(rtOptions, -1, 1, nullptr, rtMesh, nullptr);		//(the pipeline takes the arrays over)

	unsigned char pCommID[RT_COMM_ID_BYTES];
	if (rt_comm_unique_id(pCommID) != RT_OK) { std::cout << "Error while creating the device group (RCCL)\n"; return 1; }
	RankGate rtGate(rtOptions.iRanks);
	std::vector<std::thread> stdThreads;
	std::atomic<int> iFailed{ 0 };
	for (int r = 0; r < rtOptions.iRanks; r++)
		stdThreads.emplace_back([&, r]() { if (RunDevice(rtOptions, r, rtOptions.iRanks, pCommID, rtMesh, &rtGate) != 0) iFailed++; });
	for (auto& stdThread : stdThreads) stdThread.join();
	ReleaseMeshInfo(rtMesh);
	return iFailed ? 1 : 0;
}
