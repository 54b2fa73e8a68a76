// Stage sequencing of the headless ray tracer over the gpurt C-ABI. Follows the reference's
// RaytracerPipeline (src/RaytracerPipeline.cpp of the reference): which kernels run, in which order, with
// which constant-buffer values and host random seeds. Everything Direct3D-specific is gone; a stage
// fills its POD parameter struct and calls the launcher of the same shader.

#include "RaytracerPipeline.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include <zlib.h>

namespace RT::GraphicsAPI
{
	static std::random_device s_stdSeedGenerator;

	namespace
	{
		//row-vector, left-handed 4x4 helpers
		struct Matrix { float m[4][4]; };

		Matrix Transpose(const Matrix& a)
		{
			Matrix r;
			for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
			return r;
		}

		//adjugate / determinant, evaluated in binary32
		Matrix Inverse(const Matrix& a)
		{
			const float* m = &a.m[0][0];
			float c[16];
			c[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
			c[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
			c[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
			c[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
			c[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
			c[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
			c[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
			c[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
			c[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
			c[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
			c[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
			c[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
			c[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
			c[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
			c[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
			c[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
			const float fDeterminant = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
			const float fInverse = 1.0f / fDeterminant;
			Matrix r;
			for (int i = 0; i < 16; i++) (&r.m[0][0])[i] = c[i] * fInverse;
			return r;
		}

		//scalar sine and cosine by range reduction and minimax polynomials (degree 11 / 10)
		void ScalarSinCos(float* pSin, float* pCos, float fValue)
		{
			float fQuotient = 0.15915494f * fValue;
			fQuotient = (fValue >= 0.0f) ? (float)(int)(fQuotient + 0.5f) : (float)(int)(fQuotient - 0.5f);
			float y = fValue - 6.2831855f * fQuotient;
			float fSign;
			if (y > 1.5707964f) { y = 3.1415927f - y; fSign = -1.0f; }
			else if (y < -1.5707964f) { y = -3.1415927f - y; fSign = -1.0f; }
			else fSign = 1.0f;
			const float y2 = y * y;
			*pSin = (((((-2.3889859e-08f * y2 + 2.7525562e-06f) * y2 - 0.00019840874f) * y2 + 0.0083333310f) * y2 - 0.16666667f) * y2 + 1.0f) * y;
			const float p = ((((-2.6051615e-07f * y2 + 2.4760495e-05f) * y2 - 0.0013888378f) * y2 + 0.041666638f) * y2 - 0.5f) * y2 + 1.0f;
			*pCos = fSign * p;
		}

		Matrix PerspectiveFovLH(float fFovAngleY, float fAspectRatio, float fNearZ, float fFarZ)
		{
			float fSin, fCos;
			ScalarSinCos(&fSin, &fCos, 0.5f * fFovAngleY);
			const float fHeight = fCos / fSin, fWidth = fHeight / fAspectRatio, fRange = fFarZ / (fFarZ - fNearZ);
			Matrix r;
			std::memset(&r, 0, sizeof r);
			r.m[0][0] = fWidth; r.m[1][1] = fHeight; r.m[2][2] = fRange; r.m[2][3] = 1.0f; r.m[3][2] = -fRange * fNearZ;
			return r;
		}

		inline float Dot(Float3 a, Float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
		inline Float3 Cross(Float3 a, Float3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
		inline Float3 Normalize(Float3 v)
		{
			const float l = std::sqrt(Dot(v, v));
			if (l == 0.0f) return { 0.0f, 0.0f, 0.0f };
			if (std::isinf(l)) { const float q = std::nanf(""); return { q, q, q }; }
			return { v.x / l, v.y / l, v.z / l };
		}

		Matrix LookAtLH(Float3 xmEye, Float3 xmFocus, Float3 xmUp)
		{
			const Float3 r2 = Normalize({ xmFocus.x - xmEye.x, xmFocus.y - xmEye.y, xmFocus.z - xmEye.z });
			const Float3 r0 = Normalize(Cross(xmUp, r2));
			const Float3 r1 = Cross(r2, r0);
			const Float3 xmNegEye = { -xmEye.x, -xmEye.y, -xmEye.z };
			Matrix r;
			r.m[0][0] = r0.x; r.m[0][1] = r1.x; r.m[0][2] = r2.x; r.m[0][3] = 0.0f;
			r.m[1][0] = r0.y; r.m[1][1] = r1.y; r.m[1][2] = r2.y; r.m[1][3] = 0.0f;
			r.m[2][0] = r0.z; r.m[2][1] = r1.z; r.m[2][2] = r2.z; r.m[2][3] = 0.0f;
			r.m[3][0] = Dot(r0, xmNegEye); r.m[3][1] = Dot(r1, xmNegEye); r.m[3][2] = Dot(r2, xmNegEye); r.m[3][3] = 1.0f;
			return r;
		}

		uint32_t SeedOrRandom(uint32_t iSeed) { return iSeed ? iSeed : (uint32_t)s_stdSeedGenerator(); }

		//the image rows of this device: every DeviceCount-th row from row DeviceRank on
		bool RowsOfDevice(const Settings& rtSettings, rt_image_rows* rtRows)
		{
			return rt_rows_for_rank(rtSettings.WindowHeight, rtSettings.DeviceCount, rtSettings.DeviceRank, rtRows) == RT_OK;
		}
	}


	void ComputeCameraMatrices(const CameraInfo& rtCameraData, float fAspectRatio, float pInverseView[16], float pInverseProjection[16])
	{
		const Matrix xmInverseProjection = Inverse(Transpose(PerspectiveFovLH(rtCameraData.VerticalFOV, fAspectRatio, rtCameraData.NearZ, rtCameraData.FarZ)));
		const Matrix xmInverseView = Inverse(Transpose(LookAtLH(rtCameraData.Position, rtCameraData.FocusPoint, rtCameraData.UpDirection)));
		std::memcpy(pInverseView, &xmInverseView.m[0][0], 64);
		std::memcpy(pInverseProjection, &xmInverseProjection.m[0][0], 64);
	}



	//the camera ray generator class
	CameraRayGen::CameraRayGen() :
		m_rtDevice(nullptr),
		m_rtUAVDescriptorHeap(nullptr),
		m_rtInfoData(),
		m_rtRayBuffer(nullptr),
		m_rtOldRayBuffer(nullptr),
		m_rtRayPixelsBuffer(nullptr),
		m_rtRows(),
		m_stdPRNG()
	{
	}

	CameraRayGen::~CameraRayGen()
	{
		if (m_rtRayBuffer) rt_buffer_free(m_rtRayBuffer);
		if (m_rtOldRayBuffer) rt_buffer_free(m_rtOldRayBuffer);
		if (m_rtRayPixelsBuffer) rt_buffer_free(m_rtRayPixelsBuffer);
	}

	bool CameraRayGen::Initialize(rt_device* rtDevice, DescriptorHeap* rtUAVDescriptorTable, CameraInfo rtCameraData, const Settings& rtSettings)
	{
		m_rtDevice = rtDevice;
		m_rtUAVDescriptorHeap = rtUAVDescriptorTable;
		if (!m_rtDevice || !m_rtUAVDescriptorHeap) return false;
		m_stdPRNG.seed(SeedOrRandom(rtSettings.SeedRayGen));
		if (!RowsOfDevice(rtSettings, &m_rtRows)) return false;

		//ray, previous-ray and pixel-coordinate buffers (u0, u1, u2)
		const size_t iMaxRays = (size_t)rtSettings.WindowWidth * m_rtRows.num_rows * rtSettings.MaxRaysPerPixel;
		if (rt_buffer_alloc(m_rtDevice, SIZEOF_RAY * iMaxRays, &m_rtRayBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, SIZEOF_RAY * iMaxRays, &m_rtOldRayBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, SIZEOF_RAYPIXEL * iMaxRays, &m_rtRayPixelsBuffer) != RT_OK) return false;
		m_rtUAVDescriptorHeap->Slots[0] = m_rtRayBuffer;
		m_rtUAVDescriptorHeap->Slots[1] = m_rtOldRayBuffer;
		m_rtUAVDescriptorHeap->Slots[2] = m_rtRayPixelsBuffer;

		const float fAspectRatio = (float)rtSettings.WindowWidth / (float)rtSettings.WindowHeight;
		ComputeCameraMatrices(rtCameraData, fAspectRatio, m_rtInfoData.inverse_view, m_rtInfoData.inverse_projection);
		m_rtInfoData.screen_size[0] = (int32_t)rtSettings.WindowWidth;
		m_rtInfoData.screen_size[1] = (int32_t)rtSettings.WindowHeight;
		m_rtInfoData.aa_sample_spread = rtSettings.AASampleSpread;
		m_rtInfoData.dof_sample_spread = rtSettings.DOFSampleSpread;
		m_rtInfoData.max_rays_per_pixel = rtSettings.MaxRaysPerPixel;
		m_rtInfoData.rng_seed[0] = m_rtInfoData.rng_seed[1] = m_rtInfoData.rng_seed[2] = 0;
		return true;
	}

	bool CameraRayGen::Render()
	{
		//three fresh host seeds per generation
		m_rtInfoData.rng_seed[0] = (uint32_t)m_stdPRNG();
		m_rtInfoData.rng_seed[1] = (uint32_t)m_stdPRNG();
		m_rtInfoData.rng_seed[2] = (uint32_t)m_stdPRNG();
		return rt_camera_ray_generation(m_rtDevice, &m_rtInfoData, m_rtRayBuffer, m_rtOldRayBuffer, m_rtRayPixelsBuffer, &m_rtRows) == RT_OK;
	}



	//the primitive sorting class
	SortPrimitives::SortPrimitives() :
		m_rtDevice(nullptr),
		m_rtMortonCodeInfoData(),
		m_rtSortInfoData(),
		m_rtMortonCodeBuffer(nullptr),
		m_rtTempMortonCodeBuffer(nullptr),
		m_rtCodeFrequenciesBuffer(nullptr)
	{
	}

	SortPrimitives::~SortPrimitives()
	{
		if (m_rtMortonCodeBuffer) rt_buffer_free(m_rtMortonCodeBuffer);
		if (m_rtTempMortonCodeBuffer) rt_buffer_free(m_rtTempMortonCodeBuffer);
		if (m_rtCodeFrequenciesBuffer) rt_buffer_free(m_rtCodeFrequenciesBuffer);
	}

	bool SortPrimitives::Initialize(rt_device* rtDevice, uint32_t iNumPrimitives, AABB rtSceneAABB)
	{
		m_rtDevice = rtDevice;
		if (!m_rtDevice) return false;
		if (rt_buffer_alloc(m_rtDevice, 16 * (size_t)iNumPrimitives, &m_rtMortonCodeBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, 16 * (size_t)iNumPrimitives, &m_rtTempMortonCodeBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, 16 * 256, &m_rtCodeFrequenciesBuffer) != RT_OK) return false;

		m_rtMortonCodeInfoData = MortonCodeInfo{};
		m_rtMortonCodeInfoData.scene_min[0] = rtSceneAABB.Min.x; m_rtMortonCodeInfoData.scene_min[1] = rtSceneAABB.Min.y; m_rtMortonCodeInfoData.scene_min[2] = rtSceneAABB.Min.z;
		m_rtMortonCodeInfoData.scene_max[0] = rtSceneAABB.Max.x; m_rtMortonCodeInfoData.scene_max[1] = rtSceneAABB.Max.y; m_rtMortonCodeInfoData.scene_max[2] = rtSceneAABB.Max.z;
		m_rtMortonCodeInfoData.num_primitives = iNumPrimitives;
		m_rtSortInfoData.num_elements = iNumPrimitives;
		m_rtSortInfoData.sort_pass_index = 0;
		return true;
	}

	void SortPrimitives::SetSceneAABB(AABB rtSceneAABB)
	{
		m_rtMortonCodeInfoData.scene_min[0] = rtSceneAABB.Min.x; m_rtMortonCodeInfoData.scene_min[1] = rtSceneAABB.Min.y; m_rtMortonCodeInfoData.scene_min[2] = rtSceneAABB.Min.z;
		m_rtMortonCodeInfoData.scene_max[0] = rtSceneAABB.Max.x; m_rtMortonCodeInfoData.scene_max[1] = rtSceneAABB.Max.y; m_rtMortonCodeInfoData.scene_max[2] = rtSceneAABB.Max.z;
	}

	bool SortPrimitives::Sort(RaytracerMesh* rtMesh)
	{
		const rt_buffer* rtIndices = nullptr;
		const rt_buffer* rtVertices = nullptr;
		rtMesh->Bind(&rtIndices, &rtVertices);

		//Morton codes, then four 8-bit passes of count / prefix sum / scatter
		if (rt_generate_morton_codes(m_rtDevice, &m_rtMortonCodeInfoData, rtIndices, rtVertices, m_rtMortonCodeBuffer, m_rtCodeFrequenciesBuffer) != RT_OK) return false;
		for (uint32_t i = 0; i < 4; i++)
		{
			m_rtSortInfoData.sort_pass_index = i;
			if (rt_sort_count(m_rtDevice, &m_rtSortInfoData, m_rtMortonCodeBuffer, m_rtTempMortonCodeBuffer, m_rtCodeFrequenciesBuffer) != RT_OK) return false;
			if (rt_sort_prefix_sum(m_rtDevice, &m_rtSortInfoData, m_rtCodeFrequenciesBuffer) != RT_OK) return false;
			if (rt_sort_sort(m_rtDevice, &m_rtSortInfoData, m_rtMortonCodeBuffer, m_rtTempMortonCodeBuffer, m_rtCodeFrequenciesBuffer) != RT_OK) return false;
		}
		return true;
	}



	//the BVH building class
	BuildBVH::BuildBVH() :
		m_rtDevice(nullptr),
		m_iNumPrimitives(0),
		m_rtBVHInfoData(),
		m_rtBVHBuffer(nullptr),
		m_rtAccel(nullptr),
		m_bSecondHierarchy(true)
	{
	}

	BuildBVH::~BuildBVH()
	{
		if (m_rtAccel) rt_accel_destroy(m_rtAccel);
		if (m_rtBVHBuffer) rt_buffer_free(m_rtBVHBuffer);
	}

	bool BuildBVH::Initialize(rt_device* rtDevice, uint32_t iNumPrimitives)
	{
		m_rtDevice = rtDevice;
		if (!m_rtDevice) return false;
		m_iNumPrimitives = iNumPrimitives;
		//the hierarchy needs about N + 7 nodes (the reference allocates 4 N)
		const uint32_t iNumNodes = rt_bvh_node_count(iNumPrimitives);
		if (iNumNodes == 0) return false;    //fewer than 3 triangles: the level loop below would never end
		if (rt_buffer_alloc(m_rtDevice, sizeof(AABB) * (size_t)iNumNodes, &m_rtBVHBuffer) != RT_OK) return false;
		return true;
	}

	bool BuildBVH::Build(RaytracerMesh* rtMesh, rt_buffer* rtMortonCodes)
	{
		const rt_buffer* rtIndices = nullptr;
		const rt_buffer* rtVertices = nullptr;
		rtMesh->Bind(&rtIndices, &rtVertices);

		//start from the leaf schedule on every call, so that the build can be repeated each frame
		m_rtBVHInfoData.num_children = m_iNumPrimitives;
		m_rtBVHInfoData.previous_index = 1;
		m_rtBVHInfoData.current_index = ((m_iNumPrimitives + 1) / 2) + 1;
		m_rtBVHInfoData.padding = 0;

		//the leaves: one node per two sorted primitives, stored from node 1
		if (rt_bvh_build_leaves(m_rtDevice, &m_rtBVHInfoData, rtIndices, rtVertices, rtMortonCodes, m_rtBVHBuffer) != RT_OK) return false;
		m_rtBVHInfoData.num_children = (m_rtBVHInfoData.num_children + 1) / 2;

		//the rest of the tree, one level per launch, until two nodes are left
		for (; m_rtBVHInfoData.num_children != 2;)
		{
			BVHInfo rtLevel = m_rtBVHInfoData;
			m_rtBVHInfoData.num_children = (m_rtBVHInfoData.num_children + 1) / 2;
			if (rt_bvh_build(m_rtDevice, &rtLevel, m_rtBVHBuffer) != RT_OK) return false;
			m_rtBVHInfoData.previous_index = m_rtBVHInfoData.current_index;
			m_rtBVHInfoData.current_index += m_rtBVHInfoData.num_children;
		}

		//the trunk goes to node 0
		m_rtBVHInfoData.num_children = 2;
		m_rtBVHInfoData.current_index = 0;
		return rt_bvh_build(m_rtDevice, &m_rtBVHInfoData, m_rtBVHBuffer) == RT_OK;
	}

	bool BuildBVH::CreateAccel()
	{
		if (rt_accel_create(m_rtDevice, m_iNumPrimitives, &m_rtAccel) != RT_OK) return false;
		return ApplySecondHierarchy();
	}

	//a scene too large for the library's own hierarchy is still rendered (by the walk in the reference's order), but not
	//without a word: that walk is some twenty times slower
	bool BuildBVH::ApplySecondHierarchy()
	{
		const int iResult = rt_accel_set_second_hierarchy(m_rtAccel, m_bSecondHierarchy ? 1 : 0, 0xffffffffu);
		if (iResult == RT_ERR_LIMIT_EXCEEDED)
		{
			std::fprintf(stderr, "BuildBVH: %u triangles are more than the library's own hierarchy holds; tracing by the walk in the "
				"reference's order (RT_TRACE_KERNEL_QUAD), which is much slower\n", m_iNumPrimitives);
			return true;
		}
		return iResult == RT_OK;
	}

	void BuildBVH::SetSecondHierarchy(bool bEnable)
	{
		m_bSecondHierarchy = bEnable;
		if (m_rtAccel) ApplySecondHierarchy();
	}

	bool BuildBVH::BuildFusedMoved(RaytracerMesh* rtMesh, rt_buffer* rtBounds, rt_buffer* rtMortonCodes)
	{
		const rt_buffer* rtIndices = nullptr;
		const rt_buffer* rtVertices = nullptr;
		rtMesh->Bind(&rtIndices, &rtVertices);
		if (!m_rtAccel && !CreateAccel()) return false;
		if (rt_accel_pack_bounds(m_rtAccel, m_iNumPrimitives, rtIndices, rtVertices, rtBounds) != RT_OK) return false;
		return rt_accel_build_device_bounds(m_rtAccel, rtBounds, m_rtBVHBuffer, rtMortonCodes) == RT_OK;
	}

	bool BuildBVH::Refit(RaytracerMesh* rtMesh)
	{
		if (!m_rtAccel) return false;
		const rt_buffer* rtIndices = nullptr;
		const rt_buffer* rtVertices = nullptr;
		rtMesh->Bind(&rtIndices, &rtVertices);
		if (rt_accel_pack(m_rtAccel, m_iNumPrimitives, rtIndices, rtVertices) != RT_OK) return false;
		return rt_accel_refit(m_rtAccel, m_rtBVHBuffer) == RT_OK;
	}

	bool BuildBVH::BuildFused(RaytracerMesh* rtMesh, AABB rtSceneAABB, rt_buffer* rtMortonCodes)
	{
		if (!m_rtAccel)
		{
			const rt_buffer* rtIndices = nullptr;
			const rt_buffer* rtVertices = nullptr;
			rtMesh->Bind(&rtIndices, &rtVertices);
			if (!CreateAccel()) return false;
			if (rt_accel_pack(m_rtAccel, m_iNumPrimitives, rtIndices, rtVertices) != RT_OK) return false;
		}
		const float pMin[3] = { rtSceneAABB.Min.x, rtSceneAABB.Min.y, rtSceneAABB.Min.z };
		const float pMax[3] = { rtSceneAABB.Max.x, rtSceneAABB.Max.y, rtSceneAABB.Max.z };
		return rt_accel_build(m_rtAccel, pMin, pMax, m_rtBVHBuffer, rtMortonCodes) == RT_OK;
	}



	//the ray tracing class
	TraceRays::TraceRays() :
		m_rtDevice(nullptr),
		m_rtMesh(nullptr),
		m_rtTextures(nullptr),
		m_rtUAVDescriptorHeap(nullptr),
		m_rtInfoData(),
		m_rtMaterialBuffer(nullptr),
		m_rtScatteredLightBuffer(nullptr),
		m_rtEmittedLightBuffer(nullptr),
		m_rtRows(),
		m_stdPRNG(),
		m_bUseBVH(true)
	{
	}

	TraceRays::~TraceRays()
	{
		delete m_rtMesh;
		delete m_rtTextures;
		if (m_rtMaterialBuffer) rt_buffer_free(m_rtMaterialBuffer);
		if (m_rtScatteredLightBuffer) rt_buffer_free(m_rtScatteredLightBuffer);
		if (m_rtEmittedLightBuffer) rt_buffer_free(m_rtEmittedLightBuffer);
	}

	bool TraceRays::Initialize(rt_device* rtDevice, DescriptorHeap* rtUAVDescriptorTable, MeshInfo rtMeshData, const Settings& rtSettings)
	{
		m_rtDevice = rtDevice;
		m_rtUAVDescriptorHeap = rtUAVDescriptorTable;
		if (!m_rtDevice || !m_rtUAVDescriptorHeap || !rtMeshData.Materials) return false;
		m_stdPRNG.seed(SeedOrRandom(rtSettings.SeedTrace));
		m_bUseBVH = rtSettings.UseBVH;
		if (!RowsOfDevice(rtSettings, &m_rtRows)) return false;

		//materials (t2) and the two per-ray light buffers (u3, u4)
		const size_t iMaxRays = (size_t)rtSettings.WindowWidth * m_rtRows.num_rows * rtSettings.MaxRaysPerPixel;
		if (rt_buffer_alloc(m_rtDevice, sizeof(PBRMaterial) * rtMeshData.MaterialCount, &m_rtMaterialBuffer) != RT_OK) return false;
		if (rt_buffer_upload(m_rtMaterialBuffer, rtMeshData.Materials, sizeof(PBRMaterial) * rtMeshData.MaterialCount, 0) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, 16 * iMaxRays, &m_rtScatteredLightBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, 16 * iMaxRays, &m_rtEmittedLightBuffer) != RT_OK) return false;
		m_rtUAVDescriptorHeap->Slots[3] = m_rtScatteredLightBuffer;
		m_rtUAVDescriptorHeap->Slots[4] = m_rtEmittedLightBuffer;

		const uint32_t iNumIndices = (uint32_t)rtMeshData.IndexCount;

		//the texture atlas: textures are added in table order, slot 0 is the built-in white texel
		m_rtTextures = new TextureAtlas();
		for (uint64_t i = 1; i < rtMeshData.TextureNameCount; i++)
		{
			uint32_t iTextureID = 0;
			if (!(m_rtTextures->AddTexture(&iTextureID, LoadTextureFromFile(rtMeshData.TextureNames[i])))) return false;
		}
		if (!(m_rtTextures->Initialize(m_rtDevice))) return false;

		//the mesh (deletes the host arrays of rtMeshData once they are uploaded)
		m_rtMesh = new RaytracerMesh();
		if (!(m_rtMesh->Initialize(m_rtDevice, rtMeshData, !rtSettings.SharedMesh))) return false;

		m_rtInfoData.screen_dimensions[0] = (int32_t)rtSettings.WindowWidth;
		m_rtInfoData.screen_dimensions[1] = (int32_t)rtSettings.WindowHeight;
		m_rtInfoData.num_indices = iNumIndices;
		m_rtInfoData.num_rays = (uint32_t)iMaxRays;
		m_rtInfoData.max_rays_per_pixel = rtSettings.MaxRaysPerPixel;
		m_rtInfoData.rng_seed[0] = m_rtInfoData.rng_seed[1] = m_rtInfoData.rng_seed[2] = 0;
		return true;
	}

	bool TraceRays::Render(rt_buffer* rtBVH, rt_accel* rtAccel)
	{
		m_rtInfoData.rng_seed[0] = (uint32_t)m_stdPRNG();
		m_rtInfoData.rng_seed[1] = (uint32_t)m_stdPRNG();
		m_rtInfoData.rng_seed[2] = (uint32_t)m_stdPRNG();

		const rt_buffer* rtIndices = nullptr;
		const rt_buffer* rtVertices = nullptr;
		const rt_buffer* rtTextureIDs = nullptr;
		const rt_buffer* rtAtlas = nullptr;
		uint32_t iAtlasElements = 0;
		m_rtMesh->Bind(&rtIndices, &rtVertices);
		m_rtTextures->Bind(&rtTextureIDs, &rtAtlas, &iAtlasElements);
		rt_buffer** rtSlots = m_rtUAVDescriptorHeap->Slots;
		if (m_bUseBVH && rtAccel)
			return rt_trace_rays_accel(m_rtDevice, &m_rtInfoData, rtAccel, rtIndices, rtVertices, m_rtMaterialBuffer, rtTextureIDs, rtAtlas, iAtlasElements,
				rtSlots[0], rtSlots[1], rtSlots[2], rtSlots[3], rtSlots[4], rtBVH, &m_rtRows) == RT_OK;
		return rt_trace_rays(m_rtDevice, &m_rtInfoData, m_bUseBVH ? 1 : 0, rtIndices, rtVertices, m_rtMaterialBuffer, rtTextureIDs, rtAtlas, iAtlasElements,
			rtSlots[0], rtSlots[1], rtSlots[2], rtSlots[3], rtSlots[4], rtBVH, &m_rtRows) == RT_OK;
	}



	//the final image generation class
	GenerateFinalImage::GenerateFinalImage() :
		m_rtDevice(nullptr),
		m_rtUAVDescriptorHeap(nullptr),
		m_rtInfoData(),
		m_rtResultBuffer(nullptr),
		m_rtOutputTexture(nullptr)
	{
	}

	GenerateFinalImage::~GenerateFinalImage()
	{
		if (m_rtResultBuffer) rt_buffer_free(m_rtResultBuffer);
		if (m_rtOutputTexture) rt_buffer_free(m_rtOutputTexture);
	}

	bool GenerateFinalImage::Initialize(rt_device* rtDevice, DescriptorHeap* rtUAVDescriptorTable, const Settings& rtSettings)
	{
		m_rtDevice = rtDevice;
		m_rtUAVDescriptorHeap = rtUAVDescriptorTable;
		if (!m_rtDevice || !m_rtUAVDescriptorHeap) return false;
		rt_image_rows rtRows;
		if (!RowsOfDevice(rtSettings, &rtRows)) return false;
		const size_t iPixels = (size_t)rtSettings.WindowWidth * rtRows.num_rows;
		//the running mean (u5) and the tone-mapped output (u6, the reference's RGBA16F texture kept in binary32)
		if (rt_buffer_alloc(m_rtDevice, 16 * iPixels, &m_rtResultBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, 16 * iPixels, &m_rtOutputTexture) != RT_OK) return false;
		m_rtUAVDescriptorHeap->Slots[5] = m_rtResultBuffer;
		m_rtUAVDescriptorHeap->Slots[6] = m_rtOutputTexture;

		m_rtInfoData.screen_dimensions[0] = (int32_t)rtSettings.WindowWidth;
		m_rtInfoData.screen_dimensions[1] = (int32_t)rtRows.num_rows;		//the rows the buffers hold
		m_rtInfoData.max_rays_per_pixel = rtSettings.MaxRaysPerPixel;
		m_rtInfoData.num_samples = 1;
		return true;
	}

	bool GenerateFinalImage::Render(bool bApplyResults)
	{
		//bit 31 tells the kernel to commit the displayed mean; the count below it starts at 1 and is raised before its first use
		if (bApplyResults)
		{
			m_rtInfoData.num_samples |= 0x80000000;
			m_rtInfoData.num_samples++;
		}
		else
		{
			m_rtInfoData.num_samples &= 0x7fffffff;
		}
		return rt_generate_final_image(m_rtDevice, &m_rtInfoData, m_rtUAVDescriptorHeap->Slots[4], m_rtResultBuffer, m_rtOutputTexture, nullptr) == RT_OK;
	}



	//the raytracer class
	RaytracerPipeline::RaytracerPipeline() :
		m_rtDevice(nullptr),
		m_rtSettings(),
		m_rtCameraRayGen(nullptr),
		m_rtSortPrimitives(nullptr),
		m_rtBuildBVH(nullptr),
		m_rtTraceRays(nullptr),
		m_rtImageGeneration(nullptr),
		m_rtUAVDescriptorHeap(nullptr),
		m_rtSceneAABB(),
		m_bBuildBVH(true),
		m_iIteration(0),
		m_rtBaseVertexBuffer(nullptr),
		m_rtBoundsBuffer(nullptr),
		m_iFrame(0),
		m_rtRows()
	{
	}

	RaytracerPipeline::~RaytracerPipeline()
	{
		if (m_rtDevice) rt_device_sync(m_rtDevice);
		delete m_rtCameraRayGen;
		delete m_rtSortPrimitives;
		delete m_rtBuildBVH;
		delete m_rtTraceRays;
		delete m_rtImageGeneration;
		delete m_rtUAVDescriptorHeap;
		if (m_rtBaseVertexBuffer) rt_buffer_free(m_rtBaseVertexBuffer);
		if (m_rtBoundsBuffer) rt_buffer_free(m_rtBoundsBuffer);
	}

	bool RaytracerPipeline::Initialize(rt_device* rtDevice, MeshInfo rtMeshData, const Settings* rtSettings)
	{
		m_rtDevice = rtDevice;
		if (!m_rtDevice) return false;
		if (rtSettings) m_rtSettings = *rtSettings;
		if (m_rtSettings.MaxRayDepth == 0 || m_rtSettings.MaxRaysPerPixel == 0) return false;
		if (!RowsOfDevice(m_rtSettings, &m_rtRows)) return false;
		if (rt_device_set_trace_kernel(m_rtDevice, m_rtSettings.ReferenceOrderWalk ? RT_TRACE_KERNEL_QUAD : RT_TRACE_KERNEL_SPATIAL) != RT_OK)
			return false;
		if (rt_device_set_drop_missed_rays(m_rtDevice, m_rtSettings.DropMissedRays ? 1 : 0) != RT_OK) return false;
		m_rtSceneAABB = rtMeshData.SceneAABB;
		if (m_rtSettings.DynamicScene)
		{
			//the moving workload is defined on triangle soups (vertices 3i .. 3i + 2 = triangle i), which is what the loader
			//and the generator produce; it keeps the mesh as loaded beside the moved copy the stages bind
			if (!m_rtSettings.UseBVH || rtMeshData.VertexCount != rtMeshData.IndexCount) return false;
			const size_t iBytes = (size_t)rtMeshData.VertexCount * sizeof(Vertex);
			if (rt_buffer_alloc(m_rtDevice, iBytes, &m_rtBaseVertexBuffer) != RT_OK) return false;
			if (rt_buffer_upload(m_rtBaseVertexBuffer, rtMeshData.Vertices, iBytes, 0) != RT_OK) return false;
			if (rt_buffer_alloc(m_rtDevice, 32, &m_rtBoundsBuffer) != RT_OK) return false;
			m_rtSettings.RebuildBVHEveryCycle = true;
		}

		m_rtUAVDescriptorHeap = new DescriptorHeap();

		CameraInfo rtCamera{};
		rtCamera.VerticalFOV = m_rtSettings.CameraFOV;
		rtCamera.NearZ = m_rtSettings.CameraNearZ;
		rtCamera.FarZ = m_rtSettings.CameraFarZ;
		rtCamera.Position = { m_rtSettings.CameraPosition[0], m_rtSettings.CameraPosition[1], m_rtSettings.CameraPosition[2] };
		rtCamera.FocusPoint = { m_rtSettings.CameraFocusPoint[0], m_rtSettings.CameraFocusPoint[1], m_rtSettings.CameraFocusPoint[2] };
		rtCamera.UpDirection = { m_rtSettings.CameraUpDirection[0], m_rtSettings.CameraUpDirection[1], m_rtSettings.CameraUpDirection[2] };
		m_rtCameraRayGen = new CameraRayGen();
		if (!(m_rtCameraRayGen->Initialize(m_rtDevice, m_rtUAVDescriptorHeap, rtCamera, m_rtSettings))) return false;

		const uint32_t iNumPrimitives = (uint32_t)(rtMeshData.IndexCount / 3);
		if (m_rtSettings.UseBVH)
		{
			m_rtSortPrimitives = new SortPrimitives();
			if (!(m_rtSortPrimitives->Initialize(m_rtDevice, iNumPrimitives, rtMeshData.SceneAABB))) return false;
			m_rtBuildBVH = new BuildBVH();
			if (!(m_rtBuildBVH->Initialize(m_rtDevice, iNumPrimitives))) return false;
			m_rtBuildBVH->SetSecondHierarchy(!m_rtSettings.ReferenceOrderWalk);
		}

		m_rtTraceRays = new TraceRays();
		if (!(m_rtTraceRays->Initialize(m_rtDevice, m_rtUAVDescriptorHeap, rtMeshData, m_rtSettings))) return false;

		m_rtImageGeneration = new GenerateFinalImage();
		if (!(m_rtImageGeneration->Initialize(m_rtDevice, m_rtUAVDescriptorHeap, m_rtSettings))) return false;
		return true;
	}

	//DynamicScene: the triangles at their places of the next frame, in the vertex buffer the stages bind
	bool RaytracerPipeline::MoveScene()
	{
		const rt_buffer* rtIndices = nullptr;
		const rt_buffer* rtVertices = nullptr;
		m_rtTraceRays->GetMesh()->Bind(&rtIndices, &rtVertices);
		const uint32_t iNumPrimitives = (uint32_t)(m_rtTraceRays->GetMesh()->GetIndexCount() / 3);
		if (rt_synthetic_animate(m_rtDevice, iNumPrimitives, m_iFrame++, m_rtBaseVertexBuffer, const_cast<rt_buffer*>(rtVertices)) != RT_OK)
			return false;
		if (m_rtSettings.FusedBuild) return true;
		//the per-shader launchers take the scene bounds from the host, as the reference's constant buffer does
		float pBounds[8];
		if (rt_scene_bounds(m_rtDevice, iNumPrimitives, rtIndices, rtVertices, m_rtBoundsBuffer) != RT_OK) return false;
		if (rt_buffer_readback(m_rtBoundsBuffer, pBounds, sizeof(pBounds), 0) != RT_OK) return false;
		m_rtSceneAABB.Min = { pBounds[0], pBounds[1], pBounds[2] };
		m_rtSceneAABB.Max = { pBounds[4], pBounds[5], pBounds[6] };
		m_rtSortPrimitives->SetSceneAABB(m_rtSceneAABB);
		return true;
	}

	//one call = one bounce for every ray in flight
	bool RaytracerPipeline::Render()
	{
		if (m_rtSettings.UseBVH)
		{
			//the reference builds once; RebuildBVHEveryCycle repeats the build whenever new camera rays start
			if (m_bBuildBVH || (m_rtSettings.RebuildBVHEveryCycle && m_iIteration == 0))
			{
				if (m_rtSettings.DynamicScene && !MoveScene()) return false;
				if (m_rtSettings.DynamicScene && m_rtSettings.FusedBuild && m_rtSettings.RefitBVH && !m_bBuildBVH)
				{
					//every build after the first keeps the first one's order
					if (!(m_rtBuildBVH->Refit(m_rtTraceRays->GetMesh()))) return false;
				}
				else if (m_rtSettings.DynamicScene && m_rtSettings.FusedBuild)
				{
					if (!(m_rtBuildBVH->BuildFusedMoved(m_rtTraceRays->GetMesh(), m_rtBoundsBuffer, m_rtSortPrimitives->GetMortonCodes()))) return false;
				}
				else if (m_rtSettings.FusedBuild)
				{
					if (!(m_rtBuildBVH->BuildFused(m_rtTraceRays->GetMesh(), m_rtSceneAABB, m_rtSortPrimitives->GetMortonCodes()))) return false;
				}
				else
				{
					if (!(m_rtSortPrimitives->Sort(m_rtTraceRays->GetMesh()))) return false;
					if (!(m_rtBuildBVH->Build(m_rtTraceRays->GetMesh(), m_rtSortPrimitives->GetMortonCodes()))) return false;
				}
				m_bBuildBVH = false;
			}
		}

		//camera rays only on the first iteration of a cycle
		if (m_iIteration == 0)
		{
			if (!(m_rtCameraRayGen->Render())) return false;
		}
		m_iIteration++;
		if (m_iIteration == m_rtSettings.MaxRayDepth) m_iIteration = 0;

		if (!(m_rtTraceRays->Render(m_rtSettings.UseBVH ? m_rtBuildBVH->GetBVH() : nullptr,
			(m_rtSettings.UseBVH && m_rtSettings.FusedBuild) ? m_rtBuildBVH->GetAccel() : nullptr))) return false;

		//the accumulated image takes the cycle's result when the cycle is complete; in between the kernel only refreshes the
		//displayed image, which a headless caller may not want (PreviewEveryBounce)
		if (m_iIteration == 0 || m_rtSettings.PreviewEveryBounce)
		{
			if (!(m_rtImageGeneration->Render(m_iIteration == 0))) return false;
		}
		return true;
	}

	bool RaytracerPipeline::GatherOutput(rt_comm* rtComm, rt_buffer* rtImage)
	{
		if (!rtComm || rt_comm_world(rtComm) != (int)m_rtSettings.DeviceCount || rt_comm_rank(rtComm) != (int)m_rtSettings.DeviceRank) return false;
		return rt_gather_rows(rtComm, m_rtUAVDescriptorHeap->Slots[6], m_rtSettings.WindowWidth, m_rtSettings.WindowHeight, 0, rtImage) == RT_OK;
	}

	bool RaytracerPipeline::ReadOutput(float* pPixels)
	{
		const size_t iBytes = 16 * (size_t)m_rtSettings.WindowWidth * m_rtRows.num_rows;
		return rt_buffer_readback(m_rtUAVDescriptorHeap->Slots[6], pPixels, iBytes, 0) == RT_OK;
	}

	//file: "GPURTSTATE2 <width> <height> <samples> <frame>\n", the two generators' states as std::mt19937 writes them
	//(one line each: ray generation, tracing), then width * height float4 (little-endian binary32)
	bool RaytracerPipeline::SaveState(const std::string& sFileName)
	{
		if (m_iIteration != 0 || m_rtSettings.DeviceCount != 1) return false;		//only between two sample cycles, of a whole image
		//a refitted hierarchy keeps the order of the run's first build, which the file does not hold: a resumed run would refit
		//over another order and continue the uninterrupted one only up to ties between equal hit distances
		if (m_rtSettings.RefitBVH) return false;
		const size_t iPixels = (size_t)m_rtSettings.WindowWidth * m_rtSettings.WindowHeight;
		std::vector<float> stdPixels(iPixels * 4);
		if (!ReadResult(stdPixels.data())) return false;
		std::ofstream stdFile(sFileName, std::ios::binary);
		if (!stdFile) return false;
		stdFile << "GPURTSTATE2 " << m_rtSettings.WindowWidth << " " << m_rtSettings.WindowHeight << " " << m_rtImageGeneration->GetNumSamples()
			<< " " << m_iFrame << "\n";
		stdFile << m_rtCameraRayGen->GetPRNG() << "\n" << m_rtTraceRays->GetPRNG() << "\n";
		stdFile.write((const char*)stdPixels.data(), (std::streamsize)(stdPixels.size() * sizeof(float)));
		return (bool)stdFile;
	}

	bool RaytracerPipeline::LoadState(const std::string& sFileName)
	{
		if (m_iIteration != 0 || m_rtSettings.DeviceCount != 1 || m_rtSettings.RefitBVH) return false;
		std::ifstream stdFile(sFileName, std::ios::binary);
		if (!stdFile) return false;
		std::string sMagic;
		uint32_t iWidth = 0, iHeight = 0, iSamples = 0, iFrame = 0;
		stdFile >> sMagic >> iWidth >> iHeight >> iSamples >> iFrame;
		if (!stdFile || sMagic != "GPURTSTATE2" || stdFile.get() != '\n') return false;
		if (iWidth != m_rtSettings.WindowWidth || iHeight != m_rtSettings.WindowHeight || iSamples == 0) return false;
		std::mt19937 stdRayGenPRNG, stdTracePRNG;
		stdFile >> stdRayGenPRNG >> stdTracePRNG;
		if (!stdFile || stdFile.get() != '\n') return false;
		std::vector<float> stdPixels((size_t)iWidth * iHeight * 4);
		stdFile.read((char*)stdPixels.data(), (std::streamsize)(stdPixels.size() * sizeof(float)));
		if (stdFile.gcount() != (std::streamsize)(stdPixels.size() * sizeof(float))) return false;
		if (rt_buffer_upload(m_rtImageGeneration->GetResultBuffer(), stdPixels.data(), stdPixels.size() * sizeof(float), 0) != RT_OK) return false;
		m_rtImageGeneration->SetNumSamples(iSamples);
		m_rtCameraRayGen->GetPRNG() = stdRayGenPRNG;
		m_rtTraceRays->GetPRNG() = stdTracePRNG;
		m_iFrame = iFrame;
		//a moving scene is rebuilt at the restored frame before the next cycle's rays (RebuildBVHEveryCycle is on for it)
		return true;
	}

	bool RaytracerPipeline::ReadResult(float* pPixels)
	{
		const size_t iBytes = 16 * (size_t)m_rtSettings.WindowWidth * m_rtRows.num_rows;
		return rt_buffer_readback(m_rtUAVDescriptorHeap->Slots[5], pPixels, iBytes, 0) == RT_OK;
	}


	//linear -> sRGB encoding of the 8-bit sRGB back buffer the reference presents to
	uint8_t EncodeSRGB8(float fLinear)
	{
		if (!(fLinear > 0.0f)) return 0;
		if (fLinear >= 1.0f) return 255;
		const float f = fLinear <= 0.0031308f ? 12.92f * fLinear : 1.055f * std::pow(fLinear, 1.0f / 2.4f) - 0.055f;
		return (uint8_t)(f * 255.0f + 0.5f);
	}

	bool WriteImagePPM(const std::string& sFileName, const float* pPixels, uint32_t iWidth, uint32_t iHeight)
	{
		std::ofstream stdFile(sFileName, std::ios::binary);
		if (!stdFile) return false;
		stdFile << "P6\n" << iWidth << " " << iHeight << "\n255\n";
		std::vector<uint8_t> stdRow((size_t)iWidth * 3);
		for (uint32_t y = 0; y < iHeight; y++)
		{
			for (uint32_t x = 0; x < iWidth; x++)
				for (int c = 0; c < 3; c++) stdRow[3 * x + c] = EncodeSRGB8(pPixels[4 * ((size_t)y * iWidth + x) + c]);
			stdFile.write((const char*)stdRow.data(), (std::streamsize)stdRow.size());
		}
		return (bool)stdFile;
	}

	//8-bit RGB PNG (colour type 2, no interlace, filter 0 on every row, one zlib stream in one IDAT chunk)
	bool WriteImagePNG(const std::string& sFileName, const float* pPixels, uint32_t iWidth, uint32_t iHeight)
	{
		if (iWidth == 0 || iHeight == 0) return false;
		std::vector<uint8_t> stdRaw(((size_t)iWidth * 3 + 1) * iHeight);
		for (uint32_t y = 0; y < iHeight; y++)
		{
			uint8_t* pRow = stdRaw.data() + ((size_t)iWidth * 3 + 1) * y;
			pRow[0] = 0;
			for (uint32_t x = 0; x < iWidth; x++)
				for (int c = 0; c < 3; c++) pRow[1 + 3 * x + c] = EncodeSRGB8(pPixels[4 * ((size_t)y * iWidth + x) + c]);
		}
		uLongf iPackedSize = compressBound((uLong)stdRaw.size());
		std::vector<uint8_t> stdPacked(iPackedSize);
		if (compress2(stdPacked.data(), &iPackedSize, stdRaw.data(), (uLong)stdRaw.size(), 6) != Z_OK) return false;

		std::ofstream stdFile(sFileName, std::ios::binary);
		if (!stdFile) return false;
		auto Big = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
		auto Chunk = [&](const char* sType, const uint8_t* pData, uint32_t iSize)
		{
			uint8_t pWord[4];
			Big(pWord, iSize);
			stdFile.write((const char*)pWord, 4);
			stdFile.write(sType, 4);
			if (iSize) stdFile.write((const char*)pData, iSize);
			uLong iCRC = crc32(0L, (const Bytef*)sType, 4);
			if (iSize) iCRC = crc32(iCRC, pData, iSize);
			Big(pWord, (uint32_t)iCRC);
			stdFile.write((const char*)pWord, 4);
		};
		const uint8_t pSignature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
		stdFile.write((const char*)pSignature, 8);
		uint8_t pHeader[13];
		Big(pHeader, iWidth); Big(pHeader + 4, iHeight);
		pHeader[8] = 8; pHeader[9] = 2; pHeader[10] = 0; pHeader[11] = 0; pHeader[12] = 0;
		Chunk("IHDR", pHeader, 13);
		const uint8_t iIntent = 0;		//sRGB chunk: the samples are sRGB-encoded, perceptual intent
		Chunk("sRGB", &iIntent, 1);
		Chunk("IDAT", stdPacked.data(), (uint32_t)iPackedSize);
		Chunk("IEND", nullptr, 0);
		return (bool)stdFile;
	}

	bool WriteImagePFM(const std::string& sFileName, const float* pPixels, uint32_t iWidth, uint32_t iHeight)
	{
		std::ofstream stdFile(sFileName, std::ios::binary);
		if (!stdFile) return false;
		stdFile << "PF\n" << iWidth << " " << iHeight << "\n-1.0\n";
		std::vector<float> stdRow((size_t)iWidth * 3);
		for (uint32_t y = iHeight; y-- > 0;)       //PFM rows run bottom to top
		{
			for (uint32_t x = 0; x < iWidth; x++)
				for (int c = 0; c < 3; c++) stdRow[3 * x + c] = pPixels[4 * ((size_t)y * iWidth + x) + c];
			stdFile.write((const char*)stdRow.data(), (std::streamsize)(stdRow.size() * sizeof(float)));
		}
		return (bool)stdFile;
	}
}
