#pragma once
// Compile-time defaults under the reference's macro names (src/Settings.h:4-35 of the reference) and a
// run-time copy of them, so that one binary can run every benchmark configuration.
// The reference's macros carry stray semicolons that only work in the declarations they were written for;
// the names and values are the same here, the semicolons are gone. RT_CAMERA_POSITION, RT_CAMERA_FOCUS_POINT and
// RT_CAMERA_UP_DIRECTION keep the reference's brace-initialiser form (src/Settings.h:28-30 of the reference).

#include <cstdint>
#include <string>

//window size (now: size of the headless image)
#define RT_WINDOW_WIDTH 1280
#define RT_WINDOW_HEIGHT 720

//raytracing properties
#define RT_SCENE_FILENAME "assets/testscene1.obj"
#define RT_MAX_RAYS_PER_PIXEL 1
#define RT_MAX_RAY_DEPTH 3
#define RT_AA_SAMPLE_SPREAD 1.5f
#define RT_DOF_SAMPLE_SPREAD 0.0f
#define RT_USE_BVH 1
#define RT_MAX_TIME 1e30f
#define RT_MAX_SECONDS 600.0f

//camera settings
#define RT_CAMERA_FOV 1.2f
#define RT_CAMERA_NEARZ 0.01f
#define RT_CAMERA_FARZ 10000.0f

//for testscene1.obj (for testscene2.obj: position 0,0,0 focus 1,0,0; for testobject4_2.obj: position -2.5,5,-2)
#define RT_CAMERA_POSITION { -4.0259f, 2.1676f, -2.9459f }
#define RT_CAMERA_FOCUS_POINT { 0.0f, 0.0f, 0.0f }
#define RT_CAMERA_UP_DIRECTION { 0.0f, 1.0f, 0.0f }

namespace RT
{
	struct Settings
	{
		uint32_t WindowWidth = RT_WINDOW_WIDTH;
		uint32_t WindowHeight = RT_WINDOW_HEIGHT;
		std::string SceneFilename = RT_SCENE_FILENAME;
		uint32_t MaxRaysPerPixel = RT_MAX_RAYS_PER_PIXEL;
		uint32_t MaxRayDepth = RT_MAX_RAY_DEPTH;
		float AASampleSpread = RT_AA_SAMPLE_SPREAD;
		float DOFSampleSpread = RT_DOF_SAMPLE_SPREAD;
		bool UseBVH = RT_USE_BVH != 0;
		float MaxSeconds = RT_MAX_SECONDS;
		float CameraFOV = RT_CAMERA_FOV;
		float CameraNearZ = RT_CAMERA_NEARZ;
		float CameraFarZ = RT_CAMERA_FARZ;
		float CameraPosition[3] = RT_CAMERA_POSITION;
		float CameraFocusPoint[3] = RT_CAMERA_FOCUS_POINT;
		float CameraUpDirection[3] = RT_CAMERA_UP_DIRECTION;

		//additions of the headless build
		bool RebuildBVHEveryCycle = false;  //the reference builds once; true = per-frame rebuild
		bool FusedBuild = true;             //one fused LBVH build instead of one launch per reference dispatch
		bool DynamicScene = false;          //the synthetic moving workload: before every cycle the triangles are moved on the
		                                    //device (rt_synthetic_animate), the scene bounds reduced there and the tree rebuilt
		bool RefitBVH = false;              //DynamicScene with FusedBuild: the hierarchy is built once and then refitted - leaves and levels
		                                    //over the first build's sorted order (rt_accel_refit) - instead of rebuilt every cycle
		bool ReferenceOrderWalk = false;    //trace by the walk of the reference's hierarchy in the reference's order
		                                    //(RT_TRACE_KERNEL_QUAD) instead of the default on the library's own hierarchy
		bool DropMissedRays = false;        //leave rays that have missed out of the later bounces of their sample cycle instead of
		                                    //tracing them again as the reference does (same image, less work; rt_device_set_drop_missed_rays)
		bool PreviewEveryBounce = true;     //the reference shows the running image after every Render() call (one bounce); false = the
		                                    //final-image kernel runs only when a sample cycle is complete (same accumulated image)
		bool DeduplicateVertices = false;   //LoadMeshFromFile merges equal vertices (the loop the reference has switched off)
		uint32_t LoaderThreads = 0;         //threads LoadMeshFromFile parses the OBJ file with (1 = line by line; 0 = by file size and cores)
		uint32_t DeviceRank = 0;            //multi-GPU: this pipeline renders the image rows DeviceRank, DeviceRank + DeviceCount, ...
		uint32_t DeviceCount = 1;           //(one pipeline per GPU, same scene and seeds on each; RaytracerPipeline::GatherOutput assembles the image)
		bool SharedMesh = false;            //the MeshInfo handed to Initialize stays the caller's (its arrays are uploaded, not freed): several
		                                    //pipelines - one per GPU - are made from one loaded scene
		uint32_t SeedRayGen = 0;            //std::mt19937 seeds of the two host generators; 0 = std::random_device
		uint32_t SeedTrace = 0;
	};
}
