#pragma once
// Host surface of the ray tracer: the stage classes and RaytracerPipeline of the reference
// (src/RaytracerPipeline.h:20-319 of the reference) with the same class names, method names and call
// order, re-backed onto the gpurt C-ABI. A Direct3D device, command scheduler, pipeline-state objects,
// root signatures and UAV barriers have no counterpart: a stage launches its kernels on the device's
// stream, whose order is the barrier. The reference's descriptor heap (8 slots: u0..u6 and t4) is
// kept as a small table of buffer handles.

#include <random>

#include "Settings.h"
#include "gpurt.h"
#include "RaytracerMesh.h"
#include "TextureAtlas.h"


namespace RT::GraphicsAPI
{
	const unsigned int SIZEOF_RAY = 8 * 4;
	const unsigned int SIZEOF_RAYPIXEL = 4;   //one packed (x << 16 | y) word per ray

	//stand-in for the reference's UAV descriptor heap: slot i is register u<i> (slot 7: t4, the atlas)
	struct DescriptorHeap
	{
		rt_buffer* Slots[8] = {};
	};


	//the constant buffers are the C-ABI's POD structs (byte-identical to the reference's)
	typedef rt_camera_ray_gen_info CameraRayGenInfo;
	typedef rt_morton_code_info MortonCodeInfo;
	typedef rt_sort_info SortInfo;
	typedef rt_bvh_info BVHInfo;
	typedef rt_trace_rays_info TraceRaysInfo;
	typedef rt_generate_final_image_info GenerateFinalImageInfo;

	struct CameraInfo
	{
		Float3 Position;
		Float3 FocusPoint;
		Float3 UpDirection;
		float VerticalFOV;
		float NearZ;
		float FarZ;
	};

	//InverseView / InverseProjection exactly as the reference stores them: inverse(transpose(M)) of the left-handed
	//look-at and perspective matrices (row-vector convention), 16 floats row-major each
	void ComputeCameraMatrices(const CameraInfo& rtCameraData, float fAspectRatio, float pInverseView[16], float pInverseProjection[16]);


	class CameraRayGen
	{
	private:
		rt_device* m_rtDevice;
		DescriptorHeap* m_rtUAVDescriptorHeap;
		CameraRayGenInfo m_rtInfoData;
		rt_buffer* m_rtRayBuffer;
		rt_buffer* m_rtOldRayBuffer;
		rt_buffer* m_rtRayPixelsBuffer;
		rt_image_rows m_rtRows;
		std::mt19937 m_stdPRNG;

	public:
		CameraRayGen();
		~CameraRayGen();

		bool Initialize(rt_device* rtDevice, DescriptorHeap* rtUAVDescriptorTable, CameraInfo rtCameraData, const Settings& rtSettings);
		bool Render();

		const CameraRayGenInfo& GetInfo() const { return m_rtInfoData; };
		//the host generator the three seeds of every generation are drawn from (checkpoint / resume)
		std::mt19937& GetPRNG() { return m_stdPRNG; };
	};


	class SortPrimitives
	{
	private:
		rt_device* m_rtDevice;
		MortonCodeInfo m_rtMortonCodeInfoData;
		SortInfo m_rtSortInfoData;
		rt_buffer* m_rtMortonCodeBuffer;
		rt_buffer* m_rtTempMortonCodeBuffer;
		rt_buffer* m_rtCodeFrequenciesBuffer;

	public:
		SortPrimitives();
		~SortPrimitives();

		bool Initialize(rt_device* rtDevice, uint32_t iNumPrimitives, AABB rtSceneAABB);
		bool Sort(RaytracerMesh* rtMesh);

		rt_buffer* GetMortonCodes() { return m_rtMortonCodeBuffer; };
		const MortonCodeInfo& GetInfo() const { return m_rtMortonCodeInfoData; };
		//new bounds for a mesh that has moved (the reference computes them once, on load)
		void SetSceneAABB(AABB rtSceneAABB);
	};


	class BuildBVH
	{
	private:
		rt_device* m_rtDevice;
		uint32_t m_iNumPrimitives;
		BVHInfo m_rtBVHInfoData;
		rt_buffer* m_rtBVHBuffer;
		rt_accel* m_rtAccel;
		bool m_bSecondHierarchy;

		bool CreateAccel();
		bool ApplySecondHierarchy();

	public:
		BuildBVH();
		~BuildBVH();

		bool Initialize(rt_device* rtDevice, uint32_t iNumPrimitives);
		//one launch per reference dispatch (leaves, then one per level); may be called every frame
		bool Build(RaytracerMesh* rtMesh, rt_buffer* rtMortonCodes);
		//Sort + Build in one fused call (the fast path); rtMortonCodes may be null
		bool BuildFused(RaytracerMesh* rtMesh, AABB rtSceneAABB, rt_buffer* rtMortonCodes);
		//the same for a mesh that has moved on the device: gathers the triangles again, leaves their centroid bounds
		//(8 floats) in rtBounds and builds with them, without a round trip to the host
		bool BuildFusedMoved(RaytracerMesh* rtMesh, rt_buffer* rtBounds, rt_buffer* rtMortonCodes);
		//leaves and levels again over the moved vertices in the sorted order of the last fused build (rt_accel_refit);
		//false if there is no such build yet
		bool Refit(RaytracerMesh* rtMesh);

		//whether the fused builds also make the library's own hierarchy (the default traversal walks it); off for a
		//pipeline that traces in the reference's order, which never reads it
		void SetSecondHierarchy(bool bEnable);

		rt_buffer* GetBVH() { return m_rtBVHBuffer; };
		rt_accel* GetAccel() { return m_rtAccel; };
	};


	class TraceRays
	{
	private:
		rt_device* m_rtDevice;
		RaytracerMesh* m_rtMesh;
		TextureAtlas* m_rtTextures;
		DescriptorHeap* m_rtUAVDescriptorHeap;
		TraceRaysInfo m_rtInfoData;
		rt_buffer* m_rtMaterialBuffer;
		rt_buffer* m_rtScatteredLightBuffer;
		rt_buffer* m_rtEmittedLightBuffer;
		rt_image_rows m_rtRows;
		std::mt19937 m_stdPRNG;
		bool m_bUseBVH;

	public:
		TraceRays();
		~TraceRays();

		//takes ownership of the MeshInfo's arrays (RaytracerMesh::Initialize deletes them after upload)
		bool Initialize(rt_device* rtDevice, DescriptorHeap* rtUAVDescriptorTable, MeshInfo rtMeshData, const Settings& rtSettings);
		//rtAccel != null reads the accel's Morton-ordered packed triangles
		bool Render(rt_buffer* rtBVH, rt_accel* rtAccel = nullptr);

		RaytracerMesh* GetMesh() { return m_rtMesh; };
		std::mt19937& GetPRNG() { return m_stdPRNG; };
	};


	class GenerateFinalImage
	{
	private:
		rt_device* m_rtDevice;
		DescriptorHeap* m_rtUAVDescriptorHeap;
		GenerateFinalImageInfo m_rtInfoData;
		rt_buffer* m_rtResultBuffer;
		rt_buffer* m_rtOutputTexture;

	public:
		GenerateFinalImage();
		~GenerateFinalImage();

		bool Initialize(rt_device* rtDevice, DescriptorHeap* rtUAVDescriptorTable, const Settings& rtSettings);
		bool Render(bool bApplyResults = false);

		uint32_t GetNumSamples() const { return m_rtInfoData.num_samples & 0x7fffffffu; };
		//restores a sample counter saved between two cycles (checkpoint / resume)
		void SetNumSamples(uint32_t iNumSamples) { m_rtInfoData.num_samples = iNumSamples & 0x7fffffffu; };
		rt_buffer* GetResultBuffer() { return m_rtResultBuffer; };
	};


	class RaytracerPipeline
	{
	private:
		rt_device*				m_rtDevice;
		Settings				m_rtSettings;
		CameraRayGen*			m_rtCameraRayGen;
		SortPrimitives*			m_rtSortPrimitives;
		BuildBVH*				m_rtBuildBVH;
		TraceRays*				m_rtTraceRays;
		GenerateFinalImage*		m_rtImageGeneration;
		DescriptorHeap*			m_rtUAVDescriptorHeap;
		AABB					m_rtSceneAABB;
		bool					m_bBuildBVH;
		unsigned int			m_iIteration;
		rt_buffer*				m_rtBaseVertexBuffer;	//DynamicScene: the mesh as loaded
		rt_buffer*				m_rtBoundsBuffer;		//DynamicScene: centroid bounds of the moved mesh (device)
		uint32_t				m_iFrame;
		rt_image_rows			m_rtRows;				//the image rows this device renders

		bool MoveScene();

	public:
		RaytracerPipeline();
		~RaytracerPipeline();

		//rtSettings == nullptr uses the compile-time defaults of Settings.h
		bool Initialize(rt_device* rtDevice, MeshInfo rtMeshData, const Settings* rtSettings = nullptr);
		bool Render();

		//headless output instead of the window: the tone-mapped image and the accumulated mean, float4 per pixel of the
		//rows this pipeline holds (GetLocalRows() x Width; the whole image on a single device)
		bool ReadOutput(float* pPixels);
		bool ReadResult(float* pPixels);
		uint32_t GetLocalRows() const { return m_rtRows.num_rows; };
		//multi-GPU: collective over the pipelines of all devices - the tone-mapped rows travel to rank 0 (RCCL), whose
		//rtImage (Width * Height float4, device memory; null elsewhere) receives the image
		bool GatherOutput(rt_comm* rtComm, rt_buffer* rtImage);
		//checkpoint of the accumulation (the reference has none: its state lives in ResultBuffer, NumSamples and its two
		//host generators): the running-mean image, the sample counter, both std::mt19937 states and the frame of a moving
		//scene, written / restored between two sample cycles, so that k cycles + resume + k cycles = 2k cycles.
		//Refused with Settings::RefitBVH: refits keep the order of the run's first build, which the file does not hold
		bool SaveState(const std::string& sFileName);
		bool LoadState(const std::string& sFileName);
		uint32_t GetNumSamples() const { return m_rtImageGeneration ? m_rtImageGeneration->GetNumSamples() : 0; };
		//raw views for tests
		rt_buffer* GetBVH() { return m_rtBuildBVH ? m_rtBuildBVH->GetBVH() : nullptr; };
		rt_buffer* GetMortonCodes() { return m_rtSortPrimitives ? m_rtSortPrimitives->GetMortonCodes() : nullptr; };
		DescriptorHeap* GetDescriptorHeap() { return m_rtUAVDescriptorHeap; };
		const Settings& GetSettings() const { return m_rtSettings; };
	};

	//headless writers: 8-bit sRGB PPM / PNG of a tone-mapped float4 image (the encoding of the R8G8B8A8_UNORM_SRGB back
	//buffer the reference presents to, src/RenderTarget.cpp:66 of the reference), and a little-endian RGB PFM
	uint8_t EncodeSRGB8(float fLinear);
	bool WriteImagePPM(const std::string& sFileName, const float* pPixels, uint32_t iWidth, uint32_t iHeight);
	bool WriteImagePNG(const std::string& sFileName, const float* pPixels, uint32_t iWidth, uint32_t iHeight);
	bool WriteImagePFM(const std::string& sFileName, const float* pPixels, uint32_t iWidth, uint32_t iHeight);
}
