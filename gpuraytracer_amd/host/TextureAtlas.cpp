// Texture atlas packing (src/TextureAtlas.cpp:36-205 of the reference), byte for byte:
//  - slot 0 is a 1x1 white texel (8 bytes of 0xff) that every untextured material factor samples;
//  - a TextureID's Offset accumulates BYTES (16 for the first real texture, each texture rounded up
//    to 16) although the shader adds it to a TEXEL index, while the pixel data is packed tightly
//    (8 bytes for slot 0, then the textures back to back). The consequence - texture 1's texel (x, y)
//    reads image pixel x + W*y + 15, and the tail reads zeros - is the reference's behaviour and is
//    reproduced, not repaired.

#include "TextureAtlas.h"

#include <cstring>

namespace RT::GraphicsAPI
{
	TextureAtlas::TextureAtlas() :
		m_rtDevice(nullptr),
		m_stdTextureIDs(),
		m_stdTextureData(),
		m_rtTextureAtlas(nullptr),
		m_rtTextureIDs(nullptr),
		m_iBufferSize(0),
		m_iTextureCount(0)
	{
	}

	TextureAtlas::~TextureAtlas()
	{
		for (auto& rtData : m_stdTextureData) delete[] (uint8_t*)rtData.Data;
		if (m_rtTextureAtlas) rt_buffer_free(m_rtTextureAtlas);
		if (m_rtTextureIDs) rt_buffer_free(m_rtTextureIDs);
	}

	void TextureAtlas::AddDefaultTexture()
	{
		TextureID rtTextureID{ 0, 1, 1, 1 };
		TextureInfo rtTextureData{};
		rtTextureData.Data = (void*)(new uint8_t[8]);
		std::memset(rtTextureData.Data, 0xff, 8);
		rtTextureData.Width = 1;
		rtTextureData.Height = 1;
		rtTextureData.BytesPerChannel = 2;
		rtTextureData.ChannelCount = 4;
		m_stdTextureIDs.push_back(rtTextureID);
		m_stdTextureData.push_back(rtTextureData);
		m_iBufferSize += 16;
	}

	bool TextureAtlas::AddTexture(uint32_t* iTextureID, TextureInfo rtProperties)
	{
		if (iTextureID) *iTextureID = 0;
		if (m_iBufferSize == 0) AddDefaultTexture();
		if ((rtProperties.ChannelCount != 4) || (rtProperties.BytesPerChannel != 2) || !rtProperties.Data)
		{
			//the atlas owns the pixel data from the call on: a rejected texture's decoded pixels are released here
			delete[] (uint8_t*)rtProperties.Data;
			return false;
		}

		TextureID rtTextureID{};
		rtTextureID.Offset = m_iBufferSize;
		rtTextureID.Width = rtProperties.Width;
		rtTextureID.Height = rtProperties.Height;
		rtTextureID.RowPitch = rtProperties.Width;
		m_stdTextureIDs.push_back(rtTextureID);
		m_stdTextureData.push_back(rtProperties);
		m_iBufferSize += ((rtTextureID.Width * rtTextureID.Height * 8) + 15) & ~15u;
		return true;
	}

	void TextureAtlas::BuildHostImage(std::vector<uint8_t>& stdAtlasBytes, std::vector<TextureID>& stdIDs) const
	{
		if (m_iBufferSize == 0)
		{
			//nothing added yet: the image Initialize() would upload, i.e. the default slot alone
			stdAtlasBytes.assign(16, 0);
			std::memset(stdAtlasBytes.data(), 0xff, 8);
			stdIDs.assign(1, TextureID{ 0, 1, 1, 1 });
			return;
		}
		stdAtlasBytes.assign(m_iBufferSize, 0);
		size_t iOffset = 0;
		for (const TextureInfo& rtData : m_stdTextureData)
		{
			const size_t iNumBytes = (size_t)rtData.Width * rtData.Height * 8;
			std::memcpy(stdAtlasBytes.data() + iOffset, rtData.Data, iNumBytes);
			iOffset += iNumBytes;
		}
		stdIDs = m_stdTextureIDs;
	}

	bool TextureAtlas::Initialize(rt_device* rtDevice)
	{
		if (m_iBufferSize == 0) AddDefaultTexture();
		m_rtDevice = rtDevice;
		if (!m_rtDevice) return false;

		std::vector<uint8_t> stdAtlasBytes;
		std::vector<TextureID> stdIDs;
		BuildHostImage(stdAtlasBytes, stdIDs);
		if (rt_buffer_alloc(m_rtDevice, stdAtlasBytes.size(), &m_rtTextureAtlas) != RT_OK) return false;
		if (rt_buffer_upload(m_rtTextureAtlas, stdAtlasBytes.data(), stdAtlasBytes.size(), 0) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, stdIDs.size() * sizeof(TextureID), &m_rtTextureIDs) != RT_OK) return false;
		if (rt_buffer_upload(m_rtTextureIDs, stdIDs.data(), stdIDs.size() * sizeof(TextureID), 0) != RT_OK) return false;

		m_iTextureCount = (unsigned int)m_stdTextureIDs.size();
		for (auto& rtData : m_stdTextureData) delete[] (uint8_t*)rtData.Data;
		m_stdTextureIDs.clear();
		m_stdTextureData.clear();
		return true;
	}

	void TextureAtlas::Bind(const rt_buffer** rtTextureIDs, const rt_buffer** rtTextureAtlas, uint32_t* iAtlasElements) const
	{
		if (rtTextureIDs) *rtTextureIDs = m_rtTextureIDs;
		if (rtTextureAtlas) *rtTextureAtlas = m_rtTextureAtlas;
		if (iAtlasElements) *iAtlasElements = m_iBufferSize / 16;
	}
}
