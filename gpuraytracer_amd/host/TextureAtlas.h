#pragma once
// All textures of a scene in one device buffer of 16-byte elements plus a table of TextureID records
// (src/TextureAtlas.h:14-57 of the reference), uploaded through the gpurt C-ABI.

#include <vector>

#include "gpurt.h"
#include "RaytracerMesh.h"

namespace RT::GraphicsAPI
{
	struct TextureID
	{
		uint32_t Offset;
		uint32_t Width;
		uint32_t Height;
		uint32_t RowPitch;
	};
	static_assert(sizeof(TextureID) == sizeof(rt_texture_id), "host struct must keep the device layout");


	class TextureAtlas
	{
	private:

		rt_device* m_rtDevice;
		std::vector<TextureID>	m_stdTextureIDs;
		std::vector<TextureInfo> m_stdTextureData;
		rt_buffer* m_rtTextureAtlas;
		rt_buffer* m_rtTextureIDs;
		unsigned int	m_iBufferSize;
		unsigned int	m_iTextureCount;

		void AddDefaultTexture();

	public:

		TextureAtlas();
		~TextureAtlas();

		//takes ownership of rtProperties.Data (RGBA, 2 bytes per channel)
		bool AddTexture(uint32_t* iTextureID, TextureInfo rtProperties);
		bool Initialize(rt_device* rtDevice);
		//the resources a shader binds at t3 (IDs) and t4 (atlas), and the atlas length in 16-byte elements
		void Bind(const rt_buffer** rtTextureIDs, const rt_buffer** rtTextureAtlas, uint32_t* iAtlasElements) const;

		//builds the same bytes on the host (for tests and for callers that upload by themselves)
		void BuildHostImage(std::vector<uint8_t>& stdAtlasBytes, std::vector<TextureID>& stdIDs) const;

		unsigned int GetTextureCount() { return m_iTextureCount; };
		TextureID* GetTextureIDs() { return m_stdTextureIDs.data(); };
		unsigned int GetTextureIDCount() { return (unsigned int)m_stdTextureIDs.size(); };
	};
}
