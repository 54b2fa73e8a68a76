// Scene loading for the headless host: Wavefront OBJ/MTL and PNG decoding written from scratch, the
// material mapping, centroid scene bounds and normal/tangent generation of the reference's loader
// (src/RaytracerMesh.cpp:15-389 of the reference), and the mesh's device buffers over the gpurt C-ABI.
//
// The reference parses with tinyobjloader 2.0.0 and decodes with stb_image 2.27. Neither is used here;
// what is restated is their observable behaviour on the inputs the reference feeds them:
//  - reals are accumulated digit by digit in binary64 (integer part m = 10 m + d, the k-th decimal adds
//    d * 10^-k, an exponent e applies ldexp(m * 5^e, e)) and then rounded to binary32;
//  - indices are 1-based, negative ones count back from the current end; faces keep file order;
//  - a material without Kd but with map_Kd gets diffuse 0.6 (and "Kd seen" is never reset between
//    materials of one file); absent Ns is 1, absent Pr / Pm are 0;
//  - 8-bit PNG channels are widened to 16 bits as v * 257, missing alpha is 65535.

#include "RaytracerMesh.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <new>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <vector>

namespace RT::GraphicsAPI
{
	namespace
	{
		//Large arrays of the loader come straight from the kernel: blocks of this size taken from malloc by several threads
		//at once end up in per-thread heaps that grow a few pages per system call, which costs more than the parsing.
		template <typename T> struct MapAllocator
		{
			typedef T value_type;
			static constexpr size_t kDirect = 1u << 20;
			MapAllocator() = default;
			template <typename U> MapAllocator(const MapAllocator<U>&) {}
			T* allocate(size_t n)
			{
				const size_t iBytes = n * sizeof(T);
				if (iBytes < kDirect) return static_cast<T*>(::operator new(iBytes));
				void* p = mmap(nullptr, iBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
				if (p == MAP_FAILED) throw std::bad_alloc();
				(void)madvise(p, iBytes, MADV_HUGEPAGE);
				return static_cast<T*>(p);
			}
			void deallocate(T* p, size_t n)
			{
				const size_t iBytes = n * sizeof(T);
				if (iBytes < kDirect) ::operator delete(p);
				else munmap(p, iBytes);
			}
			template <typename U> bool operator==(const MapAllocator<U>&) const { return true; }
			template <typename U> bool operator!=(const MapAllocator<U>&) const { return false; }
		};
		template <typename T> using BigVector = std::vector<T, MapAllocator<T>>;

		inline Float3 Add(Float3 a, Float3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
		inline Float3 Sub(Float3 a, Float3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
		inline Float3 Scale(Float3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
		inline float Dot(Float3 a, Float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
		inline Float3 Cross(Float3 a, Float3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

		//zero for a zero-length vector, NaN for an infinite one, else v / |v|
		inline Float3 NormalizeOrZero(Float3 v)
		{
			float fLength = std::sqrt(Dot(v, v));
			if (fLength == 0.0f) return { 0.0f, 0.0f, 0.0f };
			if (std::isinf(fLength)) { float q = std::nanf(""); return { q, q, q }; }
			return { v.x / fLength, v.y / fLength, v.z / fLength };
		}

		// ------------------------------------------------------------------ PNG ----
		uint32_t ReadBE32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

		inline int Paeth(int a, int b, int c)
		{
			int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
			if (pa <= pb && pa <= pc) return a;
			if (pb <= pc) return b;
			return c;
		}

		//decodes a non-interlaced PNG (any legal bit depth and colour type) to RGBA with 16 bits per channel
		bool DecodePNG16(const std::vector<uint8_t>& stdFile, uint32_t& iWidth, uint32_t& iHeight, std::vector<uint16_t>& stdPixels, std::string& sError)
		{
			static const uint8_t pSignature[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
			if (stdFile.size() < 8 || std::memcmp(stdFile.data(), pSignature, 8) != 0) { sError = "not a PNG file"; return false; }
			size_t iPos = 8;
			uint32_t iBitDepth = 0, iColorType = 0, iInterlace = 0;
			std::vector<uint8_t> stdCompressed, stdPalette, stdTransparency;
			bool bHeader = false;
			while (iPos + 12 <= stdFile.size())
			{
				uint32_t iLength = ReadBE32(&stdFile[iPos]);
				const uint8_t* pType = &stdFile[iPos + 4];
				const uint8_t* pData = &stdFile[iPos + 8];
				if (iPos + 12 + (size_t)iLength > stdFile.size()) { sError = "truncated chunk"; return false; }
				if (!std::memcmp(pType, "IHDR", 4))
				{
					if (iLength < 13) { sError = "bad IHDR"; return false; }
					iWidth = ReadBE32(pData); iHeight = ReadBE32(pData + 4);
					iBitDepth = pData[8]; iColorType = pData[9]; iInterlace = pData[12];
					bHeader = true;
				}
				else if (!std::memcmp(pType, "PLTE", 4)) stdPalette.assign(pData, pData + iLength);
				else if (!std::memcmp(pType, "tRNS", 4)) stdTransparency.assign(pData, pData + iLength);
				else if (!std::memcmp(pType, "IDAT", 4)) stdCompressed.insert(stdCompressed.end(), pData, pData + iLength);
				else if (!std::memcmp(pType, "IEND", 4)) break;
				iPos += 12 + (size_t)iLength;
			}
			if (!bHeader || iWidth == 0 || iHeight == 0) { sError = "missing IHDR"; return false; }
			if (iInterlace != 0) { sError = "interlaced PNG is not supported"; return false; }
			if (iBitDepth != 1 && iBitDepth != 2 && iBitDepth != 4 && iBitDepth != 8 && iBitDepth != 16) { sError = "bad bit depth"; return false; }
			uint32_t iChannels = 0;
			switch (iColorType) { case 0: iChannels = 1; break; case 2: iChannels = 3; break; case 3: iChannels = 1; break;
				case 4: iChannels = 2; break; case 6: iChannels = 4; break; default: sError = "bad colour type"; return false; }
			if (iColorType == 3 && iBitDepth == 16) { sError = "bad palette depth"; return false; }
			if (iBitDepth < 8 && iColorType != 0 && iColorType != 3) { sError = "bad bit depth for this colour type"; return false; }
			//filters work on whole bytes: sub-byte pixels count as one byte
			size_t iBytesPerPixel = iBitDepth < 8 ? 1 : (size_t)iChannels * (iBitDepth / 8);
			size_t iStride = iBitDepth < 8 ? ((size_t)iWidth * iBitDepth + 7) / 8 : iBytesPerPixel * iWidth;
			std::vector<uint8_t> stdRaw((iStride + 1) * (size_t)iHeight);
			uLongf iRawSize = (uLongf)stdRaw.size();
			if (uncompress(stdRaw.data(), &iRawSize, stdCompressed.data(), (uLong)stdCompressed.size()) != Z_OK || iRawSize != stdRaw.size())
			{
				sError = "bad zlib stream"; return false;
			}
			//undo the per-row filters in place
			std::vector<uint8_t> stdImage(iStride * (size_t)iHeight);
			for (uint32_t y = 0; y < iHeight; y++)
			{
				const uint8_t* pIn = &stdRaw[(iStride + 1) * (size_t)y];
				uint8_t* pOut = &stdImage[iStride * (size_t)y];
				const uint8_t* pUp = y ? pOut - iStride : nullptr;
				const uint8_t iFilter = pIn[0];
				pIn++;
				for (size_t i = 0; i < iStride; i++)
				{
					int a = i >= iBytesPerPixel ? pOut[i - iBytesPerPixel] : 0;
					int b = pUp ? pUp[i] : 0;
					int c = (pUp && i >= iBytesPerPixel) ? pUp[i - iBytesPerPixel] : 0;
					int v = pIn[i];
					switch (iFilter) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break;
						case 4: v += Paeth(a, b, c); break; default: sError = "bad filter"; return false; }
					pOut[i] = (uint8_t)v;
				}
			}
			if (iBitDepth < 8)
			{
				//unpack 1-, 2- and 4-bit samples to one byte each; grey levels are stretched to 0..255, palette indices are not
				const uint32_t iScale = iColorType == 0 ? 255u / ((1u << iBitDepth) - 1u) : 1u;
				std::vector<uint8_t> stdUnpacked((size_t)iWidth * iHeight);
				for (uint32_t y = 0; y < iHeight; y++)
					for (uint32_t x = 0; x < iWidth; x++)
					{
						const size_t iBit = (size_t)x * iBitDepth;
						const uint8_t iByte = stdImage[iStride * (size_t)y + iBit / 8];
						const uint32_t iValue = (iByte >> (8 - iBitDepth - (iBit % 8))) & ((1u << iBitDepth) - 1u);
						stdUnpacked[(size_t)y * iWidth + x] = (uint8_t)(iValue * iScale);
					}
				stdImage.swap(stdUnpacked);
				iBitDepth = 8;
				iBytesPerPixel = 1;
				iStride = iWidth;
			}
			//expand to RGBA16
			stdPixels.assign((size_t)iWidth * iHeight * 4, 0);
			auto Sample = [&](const uint8_t* p, uint32_t c) -> uint16_t {
				if (iBitDepth == 16) return (uint16_t)((p[2 * c] << 8) | p[2 * c + 1]);
				return (uint16_t)(p[c] * 257u);
			};
			for (size_t i = 0; i < (size_t)iWidth * iHeight; i++)
			{
				const uint8_t* p = &stdImage[i * iBytesPerPixel];
				uint16_t* o = &stdPixels[i * 4];
				switch (iColorType)
				{
				case 0:
				{
					uint16_t g = Sample(p, 0); o[0] = o[1] = o[2] = g; o[3] = 65535;
					if (stdTransparency.size() >= 2)
					{
						uint16_t k = (uint16_t)((stdTransparency[0] << 8) | stdTransparency[1]);
						uint16_t r = iBitDepth == 16 ? g : (uint16_t)p[0];
						if (r == k) o[3] = 0;
					}
					break;
				}
				case 2:
				{
					o[0] = Sample(p, 0); o[1] = Sample(p, 1); o[2] = Sample(p, 2); o[3] = 65535;
					if (stdTransparency.size() >= 6)
					{
						bool bKey = true;
						for (uint32_t c = 0; c < 3; c++)
						{
							uint16_t k = (uint16_t)((stdTransparency[2 * c] << 8) | stdTransparency[2 * c + 1]);
							uint16_t r = iBitDepth == 16 ? Sample(p, c) : (uint16_t)p[c];
							bKey = bKey && (r == k);
						}
						if (bKey) o[3] = 0;
					}
					break;
				}
				case 3:
				{
					size_t e = p[0];
					if (3 * e + 2 >= stdPalette.size()) { sError = "palette index out of range"; return false; }
					o[0] = (uint16_t)(stdPalette[3 * e] * 257u); o[1] = (uint16_t)(stdPalette[3 * e + 1] * 257u); o[2] = (uint16_t)(stdPalette[3 * e + 2] * 257u);
					o[3] = e < stdTransparency.size() ? (uint16_t)(stdTransparency[e] * 257u) : 65535;
					break;
				}
				case 4: { uint16_t g = Sample(p, 0); o[0] = o[1] = o[2] = g; o[3] = Sample(p, 1); break; }
				default: o[0] = Sample(p, 0); o[1] = Sample(p, 1); o[2] = Sample(p, 2); o[3] = Sample(p, 3); break;
				}
			}
			return true;
		}

		bool ReadFile(const std::string& sFileName, std::vector<uint8_t>& stdBytes)
		{
			std::ifstream stdFile(sFileName, std::ios::binary | std::ios::ate);
			if (!stdFile) return false;
			const std::streamoff iSize = (std::streamoff)stdFile.tellg();
			if (iSize < 0) return false;
			stdBytes.resize((size_t)iSize);
			stdFile.seekg(0);
			if (iSize > 0 && !stdFile.read((char*)stdBytes.data(), iSize)) return false;
			return true;
		}

		// ------------------------------------------------------------- OBJ / MTL ----
		inline bool IsSpace(char c) { return c == ' ' || c == '\t'; }
		inline bool IsDigit(char c) { return c >= '0' && c <= '9'; }

		float NextReal(const char*& p, double fDefault = 0.0)
		{
			p += std::strspn(p, " \t");
			const char* pEnd = p + std::strcspn(p, " \t\r");
			float f = ParseReal(p, pEnd, fDefault);
			p = pEnd;
			return f;
		}

		struct ObjMaterial
		{
			std::string Name;
			float Diffuse[3] = { 0, 0, 0 }, Specular[3] = { 0, 0, 0 }, Emission[3] = { 0, 0, 0 };
			float Shininess = 1.0f, Roughness = 0.0f, Metallic = 0.0f;
			std::string DiffuseTex, RoughnessTex, SpecularTex, MetallicTex, EmissiveTex;
		};

		std::string RestOfLine(const char* p)
		{
			p += std::strspn(p, " \t");
			//option flags ("-blendu on" ...) are not interpreted: the name is the text after the last recognised option-free position
			return std::string(p);
		}

		void TrimLine(std::string& sLine)
		{
			size_t iLast = sLine.find_last_not_of(" \t");
			sLine = (iLast == std::string::npos) ? std::string() : sLine.substr(0, iLast + 1);
			while (!sLine.empty() && (sLine.back() == '\n' || sLine.back() == '\r')) sLine.pop_back();
		}

		bool LoadMTL(const std::string& sFileName, std::vector<ObjMaterial>& stdMaterials, std::unordered_map<std::string, int>& stdMaterialMap)
		{
			std::ifstream stdFile(sFileName);
			if (!stdFile) return false;
			ObjMaterial rtCurrent;
			bool bHasKd = false;    //never reset between materials (behaviour of the reference's parser)
			std::string sLine;
			auto Flush = [&]() {
				if (!rtCurrent.Name.empty())
				{
					stdMaterialMap.insert({ rtCurrent.Name, (int)stdMaterials.size() });
					stdMaterials.push_back(rtCurrent);
				}
			};
			while (std::getline(stdFile, sLine))
			{
				TrimLine(sLine);
				const char* p = sLine.c_str();
				p += std::strspn(p, " \t");
				if (*p == '\0' || *p == '#') continue;
				if (!std::strncmp(p, "newmtl", 6) && IsSpace(p[6]))
				{
					Flush();
					rtCurrent = ObjMaterial();
					rtCurrent.Name = std::string(p + 7);
					continue;
				}
				if (p[0] == 'K' && p[1] == 'd' && IsSpace(p[2])) { p += 2; for (int i = 0; i < 3; i++) rtCurrent.Diffuse[i] = NextReal(p); bHasKd = true; continue; }
				if (p[0] == 'K' && p[1] == 's' && IsSpace(p[2])) { p += 2; for (int i = 0; i < 3; i++) rtCurrent.Specular[i] = NextReal(p); continue; }
				if (p[0] == 'K' && p[1] == 'e' && IsSpace(p[2])) { p += 2; for (int i = 0; i < 3; i++) rtCurrent.Emission[i] = NextReal(p); continue; }
				if (p[0] == 'N' && p[1] == 's' && IsSpace(p[2])) { p += 2; rtCurrent.Shininess = NextReal(p); continue; }
				if (p[0] == 'P' && p[1] == 'r' && IsSpace(p[2])) { p += 2; rtCurrent.Roughness = NextReal(p); continue; }
				if (p[0] == 'P' && p[1] == 'm' && IsSpace(p[2])) { p += 2; rtCurrent.Metallic = NextReal(p); continue; }
				if (!std::strncmp(p, "map_Kd", 6) && IsSpace(p[6]))
				{
					rtCurrent.DiffuseTex = RestOfLine(p + 7);
					if (!bHasKd) { rtCurrent.Diffuse[0] = rtCurrent.Diffuse[1] = rtCurrent.Diffuse[2] = 0.6f; }
					continue;
				}
				if (!std::strncmp(p, "map_Ks", 6) && IsSpace(p[6])) { rtCurrent.SpecularTex = RestOfLine(p + 7); continue; }
				if (!std::strncmp(p, "map_Pr", 6) && IsSpace(p[6])) { rtCurrent.RoughnessTex = RestOfLine(p + 7); continue; }
				if (!std::strncmp(p, "map_Pm", 6) && IsSpace(p[6])) { rtCurrent.MetallicTex = RestOfLine(p + 7); continue; }
				if (!std::strncmp(p, "map_Ke", 6) && IsSpace(p[6])) { rtCurrent.EmissiveTex = RestOfLine(p + 7); continue; }
			}
			Flush();
			return true;
		}

		//one corner of a face as written: "v", "v/vt", "v//vn", "v/vt/vn" (the numbers as they stand, and which are there)
		struct RawCorner
		{
			int Raw[3];
			uint8_t Has;		//bit k: Raw[k] was written
		};

		RawCorner ParseFaceVertexRaw(const char*& p)
		{
			RawCorner c{ { 0, 0, 0 }, 1 };
			c.Raw[0] = std::atoi(p);
			p += std::strcspn(p, "/ \t\r");
			if (*p != '/') return c;
			p++;
			if (*p == '/')
			{
				p++;
				c.Raw[2] = std::atoi(p); c.Has |= 4;
				p += std::strcspn(p, "/ \t\r");
				return c;
			}
			c.Raw[1] = std::atoi(p); c.Has |= 2;
			p += std::strcspn(p, "/ \t\r");
			if (*p != '/') return c;
			p++;
			c.Raw[2] = std::atoi(p); c.Has |= 4;
			p += std::strcspn(p, "/ \t\r");
			return c;
		}

		//1-based or relative to the counts so far -> 0-based (-1: not written); false on a zero index
		bool FixCorner(const RawCorner& c, int iNumV, int iNumVT, int iNumVN, int iOut[3])
		{
			const int n[3] = { iNumV, iNumVT, iNumVN };
			iOut[0] = iOut[1] = iOut[2] = -1;
			for (int k = 0; k < 3; k++)
			{
				if (!(c.Has & (1 << k))) continue;
				if (c.Raw[k] > 0) iOut[k] = c.Raw[k] - 1;
				else if (c.Raw[k] == 0) return false;
				else iOut[k] = n[k] + c.Raw[k];
			}
			return true;
		}

		//returns false on a malformed or zero index
		bool ParseFaceVertex(const char*& p, int iNumV, int iNumVT, int iNumVN, int iOut[3])
		{
			const RawCorner c = ParseFaceVertexRaw(p);
			return FixCorner(c, iNumV, iNumVT, iNumVN, iOut);
		}

		void GetTexture(const std::string& sTextureName, std::unordered_map<std::string, uint32_t>& stdTextureNames)
		{
			if (stdTextureNames.find(sTextureName) == stdTextureNames.end())
			{
				uint32_t iTextureIndex = (uint32_t)stdTextureNames.size();
				stdTextureNames[sTextureName] = iTextureIndex;
			}
		}
	}


	float ParseReal(const char* pBegin, const char* pEnd, double fDefault)
	{
		double fValue = fDefault;
		const char* p = pBegin;
		do
		{
			if (p >= pEnd) break;
			double fMantissa = 0.0;
			int iExponent = 0;
			char cSign = '+', cExpSign = '+';
			bool bLeadingDot = false;
			if (*p == '+' || *p == '-') { cSign = *p; p++; if (p != pEnd && *p == '.') bLeadingDot = true; }
			else if (IsDigit(*p)) {}
			else if (*p == '.') bLeadingDot = true;
			else break;
			int iRead = 0;
			if (!bLeadingDot)
			{
				while (p != pEnd && IsDigit(*p)) { fMantissa *= 10; fMantissa += (int)(*p - '0'); p++; iRead++; }
				if (iRead == 0) break;
			}
			bool bFail = false;
			if (p != pEnd)
			{
				bool bExponentNext = false;
				if (*p == '.')
				{
					static const double pPowers[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
					p++;
					iRead = 1;
					while (p != pEnd && IsDigit(*p))
					{
						fMantissa += (int)(*p - '0') * (iRead < 8 ? pPowers[iRead] : std::pow(10.0, -iRead));
						iRead++; p++;
					}
					bExponentNext = (p != pEnd);
				}
				else if (*p == 'e' || *p == 'E') bExponentNext = true;
				if (bExponentNext && (*p == 'e' || *p == 'E'))
				{
					p++;
					if (p != pEnd && (*p == '+' || *p == '-')) { cExpSign = *p; p++; }
					else if (p != pEnd && IsDigit(*p)) {}
					else bFail = true;
					if (!bFail)
					{
						iRead = 0;
						while (p != pEnd && IsDigit(*p))
						{
							if (iExponent > 2147483647 / 10) { bFail = true; break; }
							iExponent *= 10; iExponent += (int)(*p - '0'); p++; iRead++;
						}
						iExponent *= (cExpSign == '+' ? 1 : -1);
						if (iRead == 0) bFail = true;
					}
				}
			}
			if (bFail) break;
			fValue = (cSign == '+' ? 1 : -1) * (iExponent ? std::ldexp(fMantissa * std::pow(5.0, iExponent), iExponent) : fMantissa);
		} while (false);
		return (float)fValue;
	}


	TextureInfo LoadTextureFromFile(const std::string& sFileName, int iDesiredNumChannels, bool bHighPrecision)
	{
		TextureInfo rtNothing{};
		std::vector<uint8_t> stdFile;
		std::string sError = "cannot open file";
		uint32_t iWidth = 0, iHeight = 0;
		std::vector<uint16_t> stdPixels;
		if (iDesiredNumChannels != 4) sError = "only 4 channels are supported";
		if (iDesiredNumChannels != 4 || !ReadFile(sFileName, stdFile) || !DecodePNG16(stdFile, iWidth, iHeight, stdPixels, sError))
		{
			std::cout << "Error loading textures: " << sError << " (" << sFileName << ")\n";
			return rtNothing;
		}
		TextureInfo rtTextureData{};
		const size_t iCount = (size_t)iWidth * iHeight * 4;
		if (bHighPrecision)
		{
			uint16_t* pData = new uint16_t[iCount];
			std::memcpy(pData, stdPixels.data(), iCount * 2);
			rtTextureData.Data = pData;
			rtTextureData.BytesPerChannel = 2;
		}
		else
		{
			uint8_t* pData = new uint8_t[iCount];
			for (size_t i = 0; i < iCount; i++) pData[i] = (uint8_t)(stdPixels[i] >> 8);
			rtTextureData.Data = pData;
			rtTextureData.BytesPerChannel = 1;
		}
		rtTextureData.Width = iWidth;
		rtTextureData.Height = iHeight;
		rtTextureData.ChannelCount = 4;
		std::cout << "Successfully loaded a texture with following parameters:\n Width:           " << rtTextureData.Width << "\n Height:          "
			<< rtTextureData.Height << "\n Bytes per pixel: " << (rtTextureData.BytesPerChannel * rtTextureData.ChannelCount) << "\n";
		return rtTextureData;
	}


	//per-triangle cross products summed per vertex, then normalised (src/RaytracerMesh.cpp:59-97 of the reference)
	void CalculateNormals(MeshInfo* ionMeshInfo)
	{
		const uint64_t iIndexCount = ionMeshInfo->IndexCount;
		std::vector<Float3> stdSums(iIndexCount, Float3{ 0.0f, 0.0f, 0.0f });
		for (uint64_t t = 0; t < iIndexCount / 3; t++)
		{
			const Index i1 = ionMeshInfo->Indices[3 * t], i2 = ionMeshInfo->Indices[3 * t + 1], i3 = ionMeshInfo->Indices[3 * t + 2];
			const Float3 e1 = Sub(ionMeshInfo->Vertices[i3].Position, ionMeshInfo->Vertices[i1].Position);
			const Float3 e2 = Sub(ionMeshInfo->Vertices[i2].Position, ionMeshInfo->Vertices[i1].Position);
			const Float3 n = Cross(e2, e1);
			stdSums[i1] = Add(stdSums[i1], n); stdSums[i2] = Add(stdSums[i2], n); stdSums[i3] = Add(stdSums[i3], n);
		}
		for (uint64_t i = 0; i < iIndexCount; i++)
		{
			const Index c = ionMeshInfo->Indices[i];
			ionMeshInfo->Vertices[c].Normal = NormalizeOrZero(stdSums[c]);
		}
	}

	//UV-space tangent per triangle summed per vertex, then normalised (src/RaytracerMesh.cpp:99-155 of the reference)
	void CalculateTangents(MeshInfo* ionMeshInfo)
	{
		const uint64_t iIndexCount = ionMeshInfo->IndexCount;
		std::vector<Float3> stdSums(iIndexCount, Float3{ 0.0f, 0.0f, 0.0f });
		for (uint64_t t = 0; t < iIndexCount / 3; t++)
		{
			const Index i1 = ionMeshInfo->Indices[3 * t], i2 = ionMeshInfo->Indices[3 * t + 1], i3 = ionMeshInfo->Indices[3 * t + 2];
			const Vertex& v1 = ionMeshInfo->Vertices[i1]; const Vertex& v2 = ionMeshInfo->Vertices[i2]; const Vertex& v3 = ionMeshInfo->Vertices[i3];
			const Float3 e1 = Sub(v2.Position, v1.Position), e2 = Sub(v3.Position, v1.Position);
			const float t1x = v2.UV.x - v1.UV.x, t1y = v2.UV.y - v1.UV.y, t2x = v3.UV.x - v1.UV.x, t2y = v3.UV.y - v1.UV.y;
			const float fDenominator = t1x * t2y - t1y * t2x;
			const Float3 xmNominator = Sub(Scale(e1, t2y), Scale(e2, t1y));
			const float fEpsilon = 1.192092896e-7f;
			Float3 xmTangent{ 0.0f, 0.0f, 0.0f };
			if (!(fDenominator <= fEpsilon && fDenominator >= -fEpsilon))
				xmTangent = { xmNominator.x / fDenominator, xmNominator.y / fDenominator, xmNominator.z / fDenominator };
			stdSums[i1] = Add(stdSums[i1], xmTangent); stdSums[i2] = Add(stdSums[i2], xmTangent); stdSums[i3] = Add(stdSums[i3], xmTangent);
		}
		for (uint64_t i = 0; i < iIndexCount; i++)
		{
			const Index c = ionMeshInfo->Indices[i];
			ionMeshInfo->Vertices[c].Tangent = NormalizeOrZero(stdSums[c]);
		}
	}

	//CalculateNormals (if asked) and CalculateTangents for triangles [iFrom, iTo) of a mesh whose vertices 3t, 3t + 1, 3t + 2 belong
	//to triangle t alone: the sums of the loops above have one term each (added to zero as there, which turns -0 into +0)
	static void CalculateTangentSpaceOfSoup(MeshInfo* ionMeshInfo, bool bNormals, uint64_t iFrom, uint64_t iTo)
	{
		const Float3 xmZero{ 0.0f, 0.0f, 0.0f };
		for (uint64_t t = iFrom; t < iTo; t++)
		{
			Vertex& v1 = ionMeshInfo->Vertices[3 * t]; Vertex& v2 = ionMeshInfo->Vertices[3 * t + 1]; Vertex& v3 = ionMeshInfo->Vertices[3 * t + 2];
			if (bNormals)
			{
				const Float3 n = NormalizeOrZero(Add(xmZero, Cross(Sub(v2.Position, v1.Position), Sub(v3.Position, v1.Position))));
				v1.Normal = n; v2.Normal = n; v3.Normal = n;
			}
			const Float3 e1 = Sub(v2.Position, v1.Position), e2 = Sub(v3.Position, v1.Position);
			const float t1x = v2.UV.x - v1.UV.x, t1y = v2.UV.y - v1.UV.y, t2x = v3.UV.x - v1.UV.x, t2y = v3.UV.y - v1.UV.y;
			const float fDenominator = t1x * t2y - t1y * t2x;
			const Float3 xmNominator = Sub(Scale(e1, t2y), Scale(e2, t1y));
			const float fEpsilon = 1.192092896e-7f;
			Float3 xmTangent{ 0.0f, 0.0f, 0.0f };
			if (!(fDenominator <= fEpsilon && fDenominator >= -fEpsilon))
				xmTangent = { xmNominator.x / fDenominator, xmNominator.y / fDenominator, xmNominator.z / fDenominator };
			const Float3 g = NormalizeOrZero(Add(xmZero, xmTangent));
			v1.Tangent = g; v2.Tangent = g; v3.Tangent = g;
		}
	}

	//bounds over triangle centroids, (p1 + p2 + p3) * (1/3), starting from +-FLT_MAX (src/RaytracerMesh.cpp:342-364 of the reference)
	AABB CalculateCentroidBounds(const MeshInfo& rtMesh)
	{
		const float fOneThird = 1.0f / 3.0f, fMax = 3.402823466e+38f;
		Float3 xmMin{ fMax, fMax, fMax }, xmMax{ -fMax, -fMax, -fMax };
		for (uint64_t i = 0; i + 2 < rtMesh.IndexCount; i += 3)
		{
			const Float3 c = Scale(Add(Add(rtMesh.Vertices[rtMesh.Indices[i]].Position, rtMesh.Vertices[rtMesh.Indices[i + 1]].Position),
				rtMesh.Vertices[rtMesh.Indices[i + 2]].Position), fOneThird);
			xmMin = { xmMin.x < c.x ? xmMin.x : c.x, xmMin.y < c.y ? xmMin.y : c.y, xmMin.z < c.z ? xmMin.z : c.z };
			xmMax = { xmMax.x > c.x ? xmMax.x : c.x, xmMax.y > c.y ? xmMax.y : c.y, xmMax.z > c.z ? xmMax.z : c.z };
		}
		AABB rtBounds{};
		rtBounds.Min = xmMin;
		rtBounds.Max = xmMax;
		return rtBounds;
	}


	//Removal of vertex duplicates with the semantics of the loop the reference has switched off
	//(src/RaytracerMesh.cpp:249-275 of the reference): vertex i gets the index of the first earlier unique vertex that
	//VerticesAreEqual (:157-169: position, UV, normal, material; tangents are not compared - they are computed later)
	//accepts, or becomes a new unique vertex. Equality of floats makes -0 equal +0 and a NaN unequal to everything, so the
	//hash key holds the components with -0 turned into +0, and a vertex with a NaN is always unique. Expected O(n)
	//instead of the reference's O(n^2).
	//With several threads: the vertices are dealt to the threads by their hash, every thread finds the first occurrence
	//of each of ITS vertices (equal vertices have equal hashes, so they meet in one thread), and the unique vertices are
	//then numbered in the order of their first occurrence - the numbering of the loop above.
	namespace
	{
		struct VertexKey { uint32_t w[9]; bool operator==(const VertexKey& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; } };
		struct VertexKeyHash
		{
			size_t operator()(const VertexKey& k) const
			{
				uint64_t h = 0xcbf29ce484222325ull;
				for (uint32_t x : k.w) { h ^= x; h *= 0x100000001b3ull; h ^= h >> 29; }
				return (size_t)h;
			}
		};
		//false: the vertex holds a NaN and equals nothing
		inline bool MakeVertexKey(const Vertex& v, VertexKey& k)
		{
			const float f[8] = { v.Position.x, v.Position.y, v.Position.z, v.UV.x, v.UV.y, v.Normal.x, v.Normal.y, v.Normal.z };
			bool bNaN = false;
			for (int c = 0; c < 8; c++)
			{
				const float x = f[c] == 0.0f ? 0.0f : f[c];		//-0 -> +0
				bNaN = bNaN || x != x;
				std::memcpy(&k.w[c], &x, 4);
			}
			k.w[8] = v.MaterialID;
			return !bNaN;
		}
		//rtWork(t) for t = 0 .. iThreads - 1, each on a thread of its own (0 on the caller's)
		template <typename F> void RunThreads(uint32_t iThreads, F rtWork)
		{
			std::vector<std::thread> stdThreads;
			for (uint32_t t = 1; t < iThreads; t++) stdThreads.emplace_back(rtWork, t);
			rtWork(0u);
			for (std::thread& stdThread : stdThreads) stdThread.join();
		}
	}

	static uint64_t DeduplicateVertices(const BigVector<Vertex>& stdVertices, Index* rtIndices, BigVector<Vertex>& stdUnique, uint32_t iThreads)
	{
		const size_t iCount = stdVertices.size();
		stdUnique.clear();
		if (iThreads <= 1)
		{
			std::unordered_map<VertexKey, Index, VertexKeyHash> stdFirst;
			stdFirst.reserve(iCount);
			for (size_t i = 0; i < iCount; i++)
			{
				VertexKey k;
				if (MakeVertexKey(stdVertices[i], k))
				{
					auto [stdIt, bNew] = stdFirst.try_emplace(k, (Index)stdUnique.size());
					if (!bNew) { rtIndices[i] = stdIt->second; continue; }
				}
				rtIndices[i] = (Index)stdUnique.size();
				stdUnique.push_back(stdVertices[i]);
			}
			return stdUnique.size();
		}
		//hashes (0 marks a vertex with a NaN: always unique), then per thread the first occurrences among its share
		BigVector<uint64_t> stdHash(iCount);
		RunThreads(iThreads, [&](uint32_t t) {
			for (size_t i = iCount * t / iThreads; i < iCount * (t + 1) / iThreads; i++)
			{
				VertexKey k;
				stdHash[i] = MakeVertexKey(stdVertices[i], k) ? ((uint64_t)VertexKeyHash()(k) | 1ull) : 0ull;
			}
		});
		BigVector<Index> stdFirstOf(iCount);		//index of the first vertex equal to vertex i
		RunThreads(iThreads, [&](uint32_t t) {
			std::unordered_map<VertexKey, Index, VertexKeyHash> stdFirst;
			stdFirst.reserve(iCount / iThreads + 16);
			for (size_t i = 0; i < iCount; i++)
			{
				const uint64_t h = stdHash[i];
				if (h == 0ull) { if (i % iThreads == t) stdFirstOf[i] = (Index)i; continue; }
				if ((h >> 1) % iThreads != t) continue;
				VertexKey k;
				MakeVertexKey(stdVertices[i], k);
				stdFirstOf[i] = stdFirst.try_emplace(k, (Index)i).first->second;
			}
		});
		//numbers of the unique vertices in the order of their first occurrence
		BigVector<Index> stdNumber(iCount);
		Index iUnique = 0;
		for (size_t i = 0; i < iCount; i++) if (stdFirstOf[i] == (Index)i) stdNumber[i] = iUnique++;
		stdUnique.resize(iUnique);
		RunThreads(iThreads, [&](uint32_t t) {
			for (size_t i = iCount * t / iThreads; i < iCount * (t + 1) / iThreads; i++)
			{
				rtIndices[i] = stdNumber[stdFirstOf[i]];
				if (stdFirstOf[i] == (Index)i) stdUnique[stdNumber[i]] = stdVertices[i];
			}
		});
		return iUnique;
	}


	//what both parsers hand to FinishMesh: one vertex per face corner in file order, the materials of the libraries named
	struct ParsedOBJ
	{
		BigVector<Vertex> Vertices;
		std::vector<ObjMaterial> Materials;
		bool CalculateNormals = false;
	};

	//line by line, as the reference's parser goes through the file
	static bool ParseOBJSerial(const std::string& sFileName, const std::string& sDirectory, ParsedOBJ& rtParsed)
	{
		const bool rtNothing = false;
		std::ifstream stdFile(sFileName);
		if (!stdFile)
		{
			std::cout << "Error loading scene: cannot open " << sFileName << "\n";
			return rtNothing;
		}
		std::vector<float> stdPositions, stdTexcoords, stdNormals;
		std::vector<ObjMaterial>& stdMaterials = rtParsed.Materials;
		std::unordered_map<std::string, int> stdMaterialMap;
		BigVector<Vertex>& stdVertices = rtParsed.Vertices;
		int iCurrentMaterial = -1;
		bool& bCalculateNormals = rtParsed.CalculateNormals;
		std::string sLine;
		size_t iLineNumber = 0;
		while (std::getline(stdFile, sLine))
		{
			iLineNumber++;
			TrimLine(sLine);
			const char* p = sLine.c_str();
			p += std::strspn(p, " \t");
			if (*p == '\0' || *p == '#') continue;
			if (p[0] == 'v' && IsSpace(p[1])) { p += 2; for (int i = 0; i < 3; i++) stdPositions.push_back(NextReal(p)); continue; }
			if (p[0] == 'v' && p[1] == 'n' && IsSpace(p[2])) { p += 3; for (int i = 0; i < 3; i++) stdNormals.push_back(NextReal(p)); continue; }
			if (p[0] == 'v' && p[1] == 't' && IsSpace(p[2])) { p += 3; for (int i = 0; i < 2; i++) stdTexcoords.push_back(NextReal(p)); continue; }
			if (p[0] == 'f' && IsSpace(p[1]))
			{
				p += 2;
				int iCorners = 0;
				Vertex rtCorner[3];
				std::memset(rtCorner, 0, sizeof(rtCorner));
				for (;;)
				{
					p += std::strspn(p, " \t");
					if (*p == '\0' || *p == '\r') break;
					int iIdx[3];
					if (!ParseFaceVertex(p, (int)(stdPositions.size() / 3), (int)(stdTexcoords.size() / 2), (int)(stdNormals.size() / 3), iIdx) ||
						iIdx[0] < 0 || (size_t)iIdx[0] >= stdPositions.size() / 3)
					{
						std::cout << "Error loading scene: bad face index in line " << iLineNumber << "\n";
						return rtNothing;
					}
					if (iCorners == 3)
					{
						std::cout << "Error loading scene: only triangles are supported (line " << iLineNumber << ")\n";
						return rtNothing;
					}
					Vertex& v = rtCorner[iCorners++];
					v.Position = { stdPositions[3 * iIdx[0]], stdPositions[3 * iIdx[0] + 1], stdPositions[3 * iIdx[0] + 2] };
					if (iIdx[1] >= 0 && (size_t)iIdx[1] < stdTexcoords.size() / 2) v.UV = { stdTexcoords[2 * iIdx[1]], stdTexcoords[2 * iIdx[1] + 1] };
					if (iIdx[2] >= 0 && (size_t)iIdx[2] < stdNormals.size() / 3) v.Normal = { stdNormals[3 * iIdx[2]], stdNormals[3 * iIdx[2] + 1], stdNormals[3 * iIdx[2] + 2] };
					else bCalculateNormals = true;
					v.MaterialID = (uint32_t)iCurrentMaterial;
				}
				if (iCorners != 3)
				{
					std::cout << "Error loading scene: only triangles are supported (line " << iLineNumber << ")\n";
					return rtNothing;
				}
				for (int i = 0; i < 3; i++) stdVertices.push_back(rtCorner[i]);
				continue;
			}
			if (!std::strncmp(p, "usemtl", 6) && IsSpace(p[6]))
			{
				std::string sName(p + 7);
				auto stdFound = stdMaterialMap.find(sName);
				iCurrentMaterial = stdFound == stdMaterialMap.end() ? -1 : stdFound->second;
				continue;
			}
			if (!std::strncmp(p, "mtllib", 6) && IsSpace(p[6]))
			{
				//several names may follow; the first one that loads wins
				std::stringstream stdNames(std::string(p + 7));
				std::string sName;
				while (stdNames >> sName) { if (LoadMTL(sDirectory + sName, stdMaterials, stdMaterialMap)) break; }
				continue;
			}
			//"o", "g", "s" and everything else do not change the flat triangle list
		}
		return true;
	}

	//duplicate removal, materials, bounds, normals and tangents (src/RaytracerMesh.cpp:243-389 of the reference)
	static MeshInfo FinishMesh(const std::string& sFileName, ParsedOBJ& rtParsed, bool bDeduplicate, uint32_t iThreads)
	{
		MeshInfo rtNothing{};
		BigVector<Vertex>& stdVertices = rtParsed.Vertices;
		std::vector<ObjMaterial>& stdMaterials = rtParsed.Materials;
		const bool bCalculateNormals = rtParsed.CalculateNormals;
		const uint64_t iNumVertices = stdVertices.size();
		if (iNumVertices == 0)
		{
			std::cout << "Error loading scene: no faces in " << sFileName << "\n";
			return rtNothing;
		}

		//no duplicate removal by default (the reference has it switched off): Indices[i] == i
		MeshInfo rtMesh{};
		rtMesh.IndexCount = iNumVertices;
		rtMesh.Indices = new Index[iNumVertices];
		if (bDeduplicate)
		{
			BigVector<Vertex> stdUnique;
			rtMesh.VertexCount = DeduplicateVertices(stdVertices, rtMesh.Indices, stdUnique, iThreads);
			rtMesh.Vertices = new Vertex[rtMesh.VertexCount];
			const uint64_t iUnique = rtMesh.VertexCount;
			RunThreads(iThreads, [&](uint32_t t) {
				for (uint64_t i = iUnique * t / iThreads; i < iUnique * (t + 1) / iThreads; i++) rtMesh.Vertices[i] = stdUnique[i];
			});
		}
		else
		{
			rtMesh.VertexCount = iNumVertices;
			rtMesh.Vertices = new Vertex[iNumVertices];
			RunThreads(iThreads, [&](uint32_t t) {
				for (uint64_t i = iNumVertices * t / iThreads; i < iNumVertices * (t + 1) / iThreads; i++) { rtMesh.Indices[i] = (Index)i; rtMesh.Vertices[i] = stdVertices[i]; }
			});
		}

		//materials (src/RaytracerMesh.cpp:277-321 of the reference)
		const uint64_t iNumMaterials = stdMaterials.size();
		std::unordered_map<std::string, uint32_t> stdTextureNames;
		stdTextureNames[""] = 0;
		rtMesh.MaterialCount = iNumMaterials > 0 ? iNumMaterials : 1;
		rtMesh.Materials = new PBRMaterial[rtMesh.MaterialCount];
		std::memset(rtMesh.Materials, 0, sizeof(PBRMaterial) * rtMesh.MaterialCount);
		for (uint64_t i = 0; i < iNumMaterials; i++)
		{
			const ObjMaterial& m = stdMaterials[i];
			PBRMaterial& o = rtMesh.Materials[i];
			o.Albedo = { m.Diffuse[0], m.Diffuse[1], m.Diffuse[2] };
			//shininess 1 means "no Ns given": take Pr; otherwise map Ns to a 0..1 roughness
			if (m.Shininess == 1.0f) o.Roughness = m.Roughness;
			else o.Roughness = 1.0f - (std::log2(std::fmin(std::fmax(m.Shininess, 1.0f), 1448.15f)) / 10.5f);
			o.F0Color = { m.Specular[0], m.Specular[1], m.Specular[2] };
			o.Metallic = m.Metallic;
			o.Emissive = { m.Emission[0], m.Emission[1], m.Emission[2] };
			GetTexture(m.DiffuseTex, stdTextureNames);
			GetTexture(m.RoughnessTex, stdTextureNames);
			GetTexture(m.SpecularTex, stdTextureNames);
			GetTexture(m.MetallicTex, stdTextureNames);
			GetTexture(m.EmissiveTex, stdTextureNames);
			o.AlbedoTextureID = stdTextureNames[m.DiffuseTex];
			o.RoughnessTextureID = stdTextureNames[m.RoughnessTex];
			o.F0TextureID = stdTextureNames[m.SpecularTex];
			o.MetallicTextureID = stdTextureNames[m.MetallicTex];
			o.EmissiveTextureID = stdTextureNames[m.EmissiveTex];
		}
		if (iNumMaterials < 1)
		{
			//default material: slightly glowing so that the scene is not black
			rtMesh.Materials[0].Emissive = { 0.5f, 0.5f, 0.5f };
		}
		rtMesh.TextureNameCount = stdTextureNames.size();
		rtMesh.TextureNames = new std::string[rtMesh.TextureNameCount];
		for (auto& [sTextureName, iTextureIndex] : stdTextureNames) rtMesh.TextureNames[iTextureIndex] = sTextureName;

		rtMesh.SceneAABB = CalculateCentroidBounds(rtMesh);
		if (!bDeduplicate && iThreads > 1)
		{
			//one vertex per face corner: every sum has one term, triangles are independent of each other
			const uint64_t iTriangles = rtMesh.IndexCount / 3;
			RunThreads(iThreads, [&](uint32_t t) { CalculateTangentSpaceOfSoup(&rtMesh, bCalculateNormals, iTriangles * t / iThreads, iTriangles * (t + 1) / iThreads); });
		}
		else
		{
			if (bCalculateNormals) CalculateNormals(&rtMesh);
			CalculateTangents(&rtMesh);
		}

		std::cout << "Successfully loaded the scene with:\n " << rtMesh.VertexCount << " vertices\n " << rtMesh.IndexCount << " indices\n "
			<< rtMesh.MaterialCount << " materials\n " << (rtMesh.TextureNameCount - 1) << " textures\n";
		return rtMesh;
	}


	// ---- the same file parsed by several threads ----
	//
	//The file is cut at line ends into one piece per thread. A thread parses its piece into its own arrays and keeps, per
	//face, the corners as written together with how many v / vt / vn lines of ITS piece came before: indices are 1-based or
	//relative to the counts "so far", which only exist once the pieces before are counted. Statements whose effect
	//depends on what came earlier in the file (usemtl, mtllib) are kept as events. Afterwards: counts are prefix-summed
	//over the pieces, the events replayed in file order, the first error (lowest line) reported as the serial parser
	//reports it, and the vertices made face by face in parallel. Same MeshInfo, bit for bit.
	namespace
	{
		struct FaceRecord
		{
			RawCorner Corners[3];
			uint32_t NumV, NumVT, NumVN;	//lines of the kind in this piece before the face
			uint32_t Line;					//line of the face within the piece (from 1)
			uint32_t NumCorners;			//3, or what was written (errors); more than 3: the fourth is in Extra
			uint32_t Event;					//events of this piece before the face
		};
		struct PieceEvent { bool Library; std::string Text; };
		struct Piece
		{
			const char* Begin = nullptr;
			const char* End = nullptr;
			BigVector<float> Positions, Texcoords, Normals;
			BigVector<FaceRecord> Faces;
			std::vector<std::pair<uint32_t, RawCorner>> Extra;	//(face, its fourth corner)
			std::vector<PieceEvent> Events;
			uint32_t Lines = 0;
		};

		void ParsePiece(Piece& rtPiece)
		{
			{	//lines by their first characters, so that the arrays are allocated once (a little too large: "v" needs a blank behind it)
				size_t iV = 0, iVT = 0, iVN = 0, iF = 0;
				for (const char* p = rtPiece.Begin; p < rtPiece.End;)
				{
					const char* pEol = (const char*)std::memchr(p, '\n', (size_t)(rtPiece.End - p));
					const char* pStop = pEol ? pEol : rtPiece.End;
					while (p < pStop && IsSpace(*p)) p++;
					if (p + 1 < pStop)
					{
						if (p[0] == 'v') { if (p[1] == 't') iVT++; else if (p[1] == 'n') iVN++; else iV++; }
						else if (p[0] == 'f') iF++;
					}
					p = pEol ? pEol + 1 : rtPiece.End;
				}
				rtPiece.Positions.reserve(iV * 3); rtPiece.Texcoords.reserve(iVT * 2); rtPiece.Normals.reserve(iVN * 3); rtPiece.Faces.reserve(iF);
			}
			std::string sLine;
			const char* pAt = rtPiece.Begin;
			while (pAt < rtPiece.End)
			{
				const char* pEol = (const char*)std::memchr(pAt, '\n', (size_t)(rtPiece.End - pAt));
				const char* pNext = pEol ? pEol + 1 : rtPiece.End;
				sLine.assign(pAt, pEol ? pEol : rtPiece.End);		//what std::getline delivers
				pAt = pNext;
				rtPiece.Lines++;
				TrimLine(sLine);
				const char* p = sLine.c_str();
				p += std::strspn(p, " \t");
				if (*p == '\0' || *p == '#') continue;
				if (p[0] == 'v' && IsSpace(p[1])) { p += 2; for (int i = 0; i < 3; i++) rtPiece.Positions.push_back(NextReal(p)); continue; }
				if (p[0] == 'v' && p[1] == 'n' && IsSpace(p[2])) { p += 3; for (int i = 0; i < 3; i++) rtPiece.Normals.push_back(NextReal(p)); continue; }
				if (p[0] == 'v' && p[1] == 't' && IsSpace(p[2])) { p += 3; for (int i = 0; i < 2; i++) rtPiece.Texcoords.push_back(NextReal(p)); continue; }
				if (p[0] == 'f' && IsSpace(p[1]))
				{
					p += 2;
					FaceRecord f{};
					f.NumV = (uint32_t)(rtPiece.Positions.size() / 3); f.NumVT = (uint32_t)(rtPiece.Texcoords.size() / 2);
					f.NumVN = (uint32_t)(rtPiece.Normals.size() / 3);
					f.Line = rtPiece.Lines; f.Event = (uint32_t)rtPiece.Events.size();
					for (;;)
					{
						p += std::strspn(p, " \t");
						if (*p == '\0' || *p == '\r') break;
						const RawCorner c = ParseFaceVertexRaw(p);
						if (f.NumCorners < 3) f.Corners[f.NumCorners] = c;
						else if (f.NumCorners == 3) rtPiece.Extra.push_back({ (uint32_t)rtPiece.Faces.size(), c });
						f.NumCorners++;
						if (f.NumCorners == 4) break;			//the serial parser gives up at the fourth corner, one way or the other
					}
					rtPiece.Faces.push_back(f);
					continue;
				}
				if (!std::strncmp(p, "usemtl", 6) && IsSpace(p[6])) { rtPiece.Events.push_back({ false, std::string(p + 7) }); continue; }
				if (!std::strncmp(p, "mtllib", 6) && IsSpace(p[6])) { rtPiece.Events.push_back({ true, std::string(p + 7) }); continue; }
			}
		}

	}

	static bool ParseOBJParallel(const std::string& sFileName, const std::string& sDirectory, uint32_t iThreads, ParsedOBJ& rtParsed)
	{
		//the file is mapped, not read: its pages are the page cache's
		std::vector<uint8_t> stdBytes;
		const char* pMapped = nullptr;
		size_t iMappedSize = 0;
		{
			const int iFile = open(sFileName.c_str(), O_RDONLY);
			struct stat rtStat;
			if (iFile >= 0 && fstat(iFile, &rtStat) == 0 && rtStat.st_size > 0)
			{
				void* p = mmap(nullptr, (size_t)rtStat.st_size, PROT_READ, MAP_PRIVATE, iFile, 0);
				if (p != MAP_FAILED) { pMapped = (const char*)p; iMappedSize = (size_t)rtStat.st_size; }
			}
			if (iFile >= 0) close(iFile);
		}
		struct Unmap { const char* p; size_t n; ~Unmap() { if (p) munmap((void*)p, n); } } rtUnmap{ pMapped, iMappedSize };
		if (!pMapped && !ReadFile(sFileName, stdBytes))
		{
			std::cout << "Error loading scene: cannot open " << sFileName << "\n";
			return false;
		}
		const char* pFile = pMapped ? pMapped : (const char*)stdBytes.data();
		const size_t iSize = pMapped ? iMappedSize : stdBytes.size();
		//pieces of about equal size that end behind a line end
		std::vector<Piece> stdPieces(iThreads);
		size_t iFrom = 0;
		for (uint32_t t = 0; t < iThreads; t++)
		{
			size_t iTo = t + 1 == iThreads ? iSize : std::max(iFrom, iSize * (t + 1) / iThreads);
			while (iTo < iSize && iTo > 0 && pFile[iTo - 1] != '\n') iTo++;
			stdPieces[t].Begin = pFile + iFrom; stdPieces[t].End = pFile + iTo;
			iFrom = iTo;
		}
		RunThreads(iThreads, [&](uint32_t t) { ParsePiece(stdPieces[t]); });

		//what came before every piece
		std::vector<uint64_t> stdV(iThreads + 1, 0), stdVT(iThreads + 1, 0), stdVN(iThreads + 1, 0), stdF(iThreads + 1, 0), stdL(iThreads + 1, 0);
		for (uint32_t t = 0; t < iThreads; t++)
		{
			stdV[t + 1] = stdV[t] + stdPieces[t].Positions.size() / 3; stdVT[t + 1] = stdVT[t] + stdPieces[t].Texcoords.size() / 2;
			stdVN[t + 1] = stdVN[t] + stdPieces[t].Normals.size() / 3; stdF[t + 1] = stdF[t] + stdPieces[t].Faces.size();
			stdL[t + 1] = stdL[t] + stdPieces[t].Lines;
		}
		BigVector<float> stdPositions(stdV[iThreads] * 3), stdTexcoords(stdVT[iThreads] * 2), stdNormals(stdVN[iThreads] * 3);
		RunThreads(iThreads, [&](uint32_t t) {
			const Piece& c = stdPieces[t];
			if (!c.Positions.empty()) std::memcpy(&stdPositions[stdV[t] * 3], c.Positions.data(), c.Positions.size() * sizeof(float));
			if (!c.Texcoords.empty()) std::memcpy(&stdTexcoords[stdVT[t] * 2], c.Texcoords.data(), c.Texcoords.size() * sizeof(float));
			if (!c.Normals.empty()) std::memcpy(&stdNormals[stdVN[t] * 3], c.Normals.data(), c.Normals.size() * sizeof(float));
		});

		//vertices face by face; a piece remembers its first bad face: (line, 0 = bad index, 1 = not a triangle)
		const uint64_t iNumFaces = stdF[iThreads];
		rtParsed.Vertices.assign(iNumFaces * 3, Vertex{});
		std::vector<std::pair<uint64_t, int>> stdFirstError(iThreads, { ~0ull, 0 });
		std::vector<uint8_t> stdNeedNormals(iThreads, 0);
		RunThreads(iThreads, [&](uint32_t t) {
			const Piece& c = stdPieces[t];
			size_t iExtra = 0;
			for (size_t k = 0; k < c.Faces.size(); k++)
			{
				const FaceRecord& f = c.Faces[k];
				const int iNumV = (int)(stdV[t] + f.NumV), iNumVT = (int)(stdVT[t] + f.NumVT), iNumVN = (int)(stdVN[t] + f.NumVN);
				Vertex* rtCorner = &rtParsed.Vertices[(stdF[t] + k) * 3];
				int iError = -1;
				for (uint32_t j = 0; j < f.NumCorners && iError < 0; j++)
				{
					RawCorner rtRaw;
					if (j < 3) rtRaw = f.Corners[j];
					else { while (c.Extra[iExtra].first != k) iExtra++; rtRaw = c.Extra[iExtra].second; }
					int iIdx[3];
					if (!FixCorner(rtRaw, iNumV, iNumVT, iNumVN, iIdx) || iIdx[0] < 0 || iIdx[0] >= iNumV) { iError = 0; break; }
					if (j == 3) { iError = 1; break; }
					Vertex& v = rtCorner[j];
					v.Position = { stdPositions[3 * (size_t)iIdx[0]], stdPositions[3 * (size_t)iIdx[0] + 1], stdPositions[3 * (size_t)iIdx[0] + 2] };
					if (iIdx[1] >= 0 && iIdx[1] < iNumVT) v.UV = { stdTexcoords[2 * (size_t)iIdx[1]], stdTexcoords[2 * (size_t)iIdx[1] + 1] };
					if (iIdx[2] >= 0 && iIdx[2] < iNumVN) v.Normal = { stdNormals[3 * (size_t)iIdx[2]], stdNormals[3 * (size_t)iIdx[2] + 1], stdNormals[3 * (size_t)iIdx[2] + 2] };
					else stdNeedNormals[t] = 1;
				}
				if (iError < 0 && f.NumCorners != 3) iError = 1;
				if (iError >= 0) { stdFirstError[t] = { stdL[t] + f.Line, iError }; return; }
			}
		});
		for (uint32_t t = 0; t < iThreads; t++)
		{
			if (stdFirstError[t].first == ~0ull) continue;
			if (stdFirstError[t].second == 0) std::cout << "Error loading scene: bad face index in line " << stdFirstError[t].first << "\n";
			else std::cout << "Error loading scene: only triangles are supported (line " << stdFirstError[t].first << ")\n";
			return false;
		}
		for (uint32_t t = 0; t < iThreads; t++) rtParsed.CalculateNormals = rtParsed.CalculateNormals || stdNeedNormals[t] != 0;

		//events in file order: the material every face is written with
		std::unordered_map<std::string, int> stdMaterialMap;
		int iCurrentMaterial = -1;
		std::vector<std::vector<int>> stdMaterialAfter(iThreads);		//[piece][e] = material after e events of the piece
		for (uint32_t t = 0; t < iThreads; t++)
		{
			stdMaterialAfter[t].push_back(iCurrentMaterial);
			for (const PieceEvent& e : stdPieces[t].Events)
			{
				if (e.Library)
				{
					std::stringstream stdNames(e.Text);
					std::string sName;
					while (stdNames >> sName) { if (LoadMTL(sDirectory + sName, rtParsed.Materials, stdMaterialMap)) break; }
				}
				else
				{
					auto stdFound = stdMaterialMap.find(e.Text);
					iCurrentMaterial = stdFound == stdMaterialMap.end() ? -1 : stdFound->second;
				}
				stdMaterialAfter[t].push_back(iCurrentMaterial);
			}
		}
		RunThreads(iThreads, [&](uint32_t t) {
			const Piece& c = stdPieces[t];
			for (size_t k = 0; k < c.Faces.size(); k++)
			{
				const uint32_t iMaterial = (uint32_t)stdMaterialAfter[t][c.Faces[k].Event];
				Vertex* rtCorner = &rtParsed.Vertices[(stdF[t] + k) * 3];
				rtCorner[0].MaterialID = rtCorner[1].MaterialID = rtCorner[2].MaterialID = iMaterial;
			}
		});
		return true;
	}

	MeshInfo LoadMeshFromFile(const std::string& sFileName, bool bDeduplicate, uint32_t iThreads)
	{
		//the material library is looked up next to the OBJ file
		std::string sDirectory;
		{
			size_t iSlash = sFileName.find_last_of("/\\");
			if (iSlash != std::string::npos) sDirectory = sFileName.substr(0, iSlash + 1);
		}
		if (iThreads == 0)
		{
			//small files are not worth the threads; the share of a GPU box's cores that one GPU gets is 16
			std::ifstream stdProbe(sFileName, std::ios::binary | std::ios::ate);
			const std::streamoff iBytes = stdProbe ? (std::streamoff)stdProbe.tellg() : 0;
			const uint32_t iCores = std::thread::hardware_concurrency();
			iThreads = iBytes < (1 << 20) ? 1u : std::min(16u, std::max(1u, iCores));
		}
		ParsedOBJ rtParsed;
		const bool bParsed = iThreads == 1 ? ParseOBJSerial(sFileName, sDirectory, rtParsed)
		                                    : ParseOBJParallel(sFileName, sDirectory, std::min(iThreads, 256u), rtParsed);
		if (!bParsed) return MeshInfo{};
		return FinishMesh(sFileName, rtParsed, bDeduplicate, iThreads);
	}


	void ReleaseMeshInfo(MeshInfo& rtMesh)
	{
		delete[] rtMesh.Indices;
		delete[] rtMesh.Vertices;
		delete[] rtMesh.Materials;
		delete[] rtMesh.TextureNames;
		rtMesh = MeshInfo{};
	}


	RaytracerMesh::RaytracerMesh() :
		m_rtDevice(nullptr),
		m_rtIndexBuffer(nullptr),
		m_rtVertexBuffer(nullptr),
		m_rtMesh()
	{
	}

	RaytracerMesh::~RaytracerMesh()
	{
		Release();
	}

	bool RaytracerMesh::Initialize(rt_device* rtDevice, MeshInfo rtMesh, bool bTakeOwnership)
	{
		m_rtDevice = rtDevice;
		if (!m_rtDevice || !rtMesh.Indices || !rtMesh.Vertices) return false;
		m_rtMesh = rtMesh;
		const size_t iIndexBytes = rtMesh.IndexCount * sizeof(Index), iVertexBytes = rtMesh.VertexCount * sizeof(Vertex);
		if (rt_buffer_alloc(m_rtDevice, iIndexBytes, &m_rtIndexBuffer) != RT_OK) return false;
		if (rt_buffer_alloc(m_rtDevice, iVertexBytes, &m_rtVertexBuffer) != RT_OK) return false;
		if (rt_buffer_upload(m_rtIndexBuffer, rtMesh.Indices, iIndexBytes, 0) != RT_OK) return false;
		if (rt_buffer_upload(m_rtVertexBuffer, rtMesh.Vertices, iVertexBytes, 0) != RT_OK) return false;

		//the mesh data now lives on the gpu
		if (bTakeOwnership)
		{
			delete[] rtMesh.Indices;
			delete[] rtMesh.Vertices;
			delete[] rtMesh.Materials;
		}
		m_rtMesh.Indices = nullptr;
		m_rtMesh.Vertices = nullptr;
		m_rtMesh.Materials = nullptr;
		return true;
	}

	void RaytracerMesh::Bind(const rt_buffer** rtIndexBuffer, const rt_buffer** rtVertexBuffer) const
	{
		if (rtIndexBuffer) *rtIndexBuffer = m_rtIndexBuffer;
		if (rtVertexBuffer) *rtVertexBuffer = m_rtVertexBuffer;
	}

	void RaytracerMesh::Release()
	{
		if (m_rtIndexBuffer) { rt_buffer_free(m_rtIndexBuffer); m_rtIndexBuffer = nullptr; }
		if (m_rtVertexBuffer) { rt_buffer_free(m_rtVertexBuffer); m_rtVertexBuffer = nullptr; }
	}
}
