// PNG reader for image textures.
//
// The reference loads texture files with lodepng::decode( image, w, h, filename ) (image.cpp:13-40),
// whose default output is 8-bit RGBA whatever the file holds, and keeps r, g, b of every pixel.
// This reader produces those r, g, b bytes: every colour type (grey, RGB, palette, grey + alpha,
// RGBA), every bit depth (1, 2, 4, 8, 16) and Adam7 interlacing, with lodepng's conversion to
// 8 bits (16-bit samples keep their high byte; grey samples below 8 bits are scaled by
// 255 / (2^depth - 1) with integer division; palette indices are looked up).  Checksums (CRC-32 of
// chunks, Adler-32 of the zlib stream) are not verified, as lodepng's decoder can be told to skip
// them; a corrupt stream fails in inflate or in the size check of the decoded image.
#include "host_math.hpp"

#include <cstdio>
#include <cstring>

namespace mrt {

namespace {

struct BitReader {
    const unsigned char* p;
    size_t n, pos = 0; // pos in bits
    bool bad = false;
    // the next 32 - 7 or more bits, least significant first; zeros beyond the end
    uint32_t peek() const {
        const size_t byte = pos >> 3;
        uint64_t v        = 0;
        for ( size_t i = 0; i < 5 && byte + i < n; i++ )
            v |= (uint64_t)p[byte + i] << ( 8 * i );
        return (uint32_t)( v >> ( pos & 7 ) );
    }
    void skip( int k ) {
        pos += (size_t)k;
        if ( pos > n * 8 ) bad = true;
    }
    uint32_t bits( int k ) { // k <= 16
        const uint32_t v = peek() & ( ( 1u << k ) - 1u );
        skip( k );
        return bad ? 0u : v;
    }
};

// canonical Huffman code: counts per length + symbols ordered by (length, symbol)
struct Huffman {
    uint16_t count[16];
    std::vector<uint16_t> symbol;
    bool build( const uint8_t* lengths, int n ) {
        std::memset( count, 0, sizeof( count ) );
        for ( int i = 0; i < n; i++ )
            count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for ( int len = 1; len < 16; len++ ) {
            left <<= 1;
            left -= count[len];
            if ( left < 0 ) return false; // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for ( int len = 1; len < 15; len++ )
            offs[len + 1] = offs[len] + count[len];
        symbol.assign( n, 0 );
        for ( int i = 0; i < n; i++ )
            if ( lengths[i] ) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode( BitReader& br ) const {
        uint32_t window = br.peek();
        int code = 0, first = 0, index = 0;
        for ( int len = 1; len < 16; len++ ) {
            code |= (int)( window & 1u );
            window >>= 1;
            int c = count[len];
            if ( code - c < first ) {
                br.skip( len );
                return br.bad ? -1 : symbol[index + ( code - first )];
            }
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// the decoded stream may not grow beyond this (set by the caller from the picture's header: a file
// cannot make the reader allocate more than its own header accounts for)
thread_local size_t g_inflate_limit = ~(size_t)0;

bool inflate_codes( BitReader& br, std::vector<unsigned char>& out, const Huffman& lit, const Huffman& dist ) {
    static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint16_t lext[29]  = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const uint16_t dext[30]  = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    for ( ;; ) {
        int sym = lit.decode( br );
        if ( sym < 0 ) return false;
        if ( sym < 256 ) {
            if ( out.size() >= g_inflate_limit ) return false;
            out.push_back( (unsigned char)sym );
        }
        else if ( sym == 256 ) return true;
        else {
            sym -= 257;
            if ( sym >= 29 ) return false;
            size_t len = lbase[sym] + br.bits( lext[sym] );
            int ds     = dist.decode( br );
            if ( ds < 0 || ds >= 30 ) return false;
            size_t d = dbase[ds] + br.bits( dext[ds] );
            if ( br.bad || d > out.size() || len > g_inflate_limit - out.size() ) return false;
            size_t from = out.size() - d;
            for ( size_t i = 0; i < len; i++ )
                out.push_back( out[from + i] );
        }
    }
}

// zlib stream (RFC 1950) holding deflate blocks (RFC 1951)
bool zlib_inflate( const unsigned char* z, size_t n, std::vector<unsigned char>& out ) {
    if ( n < 2 || ( z[0] & 0x0f ) != 8 || ( ( z[0] << 8 | z[1] ) % 31 ) != 0 || ( z[1] & 0x20 ) ) return false;
    BitReader br { z + 2, n - 2 };
    Huffman fixed_lit, fixed_dist;
    {
        uint8_t l[288];
        for ( int i = 0; i < 144; i++ ) l[i] = 8;
        for ( int i = 144; i < 256; i++ ) l[i] = 9;
        for ( int i = 256; i < 280; i++ ) l[i] = 7;
        for ( int i = 280; i < 288; i++ ) l[i] = 8;
        fixed_lit.build( l, 288 );
        uint8_t dl[30];
        for ( int i = 0; i < 30; i++ ) dl[i] = 5;
        fixed_dist.build( dl, 30 );
    }
    for ( ;; ) {
        uint32_t last = br.bits( 1 ), type = br.bits( 2 );
        if ( br.bad ) return false;
        if ( type == 0 ) {
            br.pos       = ( br.pos + 7 ) & ~(size_t)7;
            size_t byte  = br.pos >> 3;
            if ( byte + 4 > br.n ) return false;
            uint32_t len = br.p[byte] | ( br.p[byte + 1] << 8 ), nlen = br.p[byte + 2] | ( br.p[byte + 3] << 8 );
            if ( ( len ^ 0xffffu ) != nlen || byte + 4 + len > br.n || len > g_inflate_limit - out.size() ) return false;
            out.insert( out.end(), br.p + byte + 4, br.p + byte + 4 + len );
            br.pos = ( byte + 4 + len ) << 3;
        }
        else if ( type == 1 ) {
            if ( !inflate_codes( br, out, fixed_lit, fixed_dist ) ) return false;
        }
        else if ( type == 2 ) {
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            int nlen = (int)br.bits( 5 ) + 257, ndist = (int)br.bits( 5 ) + 1, ncode = (int)br.bits( 4 ) + 4;
            if ( br.bad || nlen > 286 || ndist > 30 ) return false;
            uint8_t lengths[320];
            std::memset( lengths, 0, sizeof( lengths ) );
            for ( int i = 0; i < ncode; i++ )
                lengths[order[i]] = (uint8_t)br.bits( 3 );
            Huffman lencode;
            if ( !lencode.build( lengths, 19 ) ) return false;
            int idx = 0;
            uint8_t ll[320];
            std::memset( ll, 0, sizeof( ll ) );
            while ( idx < nlen + ndist ) {
                int sym = lencode.decode( br );
                if ( sym < 0 ) return false;
                if ( sym < 16 ) ll[idx++] = (uint8_t)sym;
                else {
                    int rep     = 0;
                    uint8_t val = 0;
                    if ( sym == 16 ) {
                        if ( idx == 0 ) return false;
                        val = ll[idx - 1];
                        rep = 3 + (int)br.bits( 2 );
                    }
                    else if ( sym == 17 ) rep = 3 + (int)br.bits( 3 );
                    else rep = 11 + (int)br.bits( 7 );
                    if ( br.bad || idx + rep > nlen + ndist ) return false;
                    while ( rep-- )
                        ll[idx++] = val;
                }
            }
            if ( ll[256] == 0 ) return false;
            Huffman lit, dist;
            if ( !lit.build( ll, nlen ) || !dist.build( ll + nlen, ndist ) ) return false;
            if ( !inflate_codes( br, out, lit, dist ) ) return false;
        }
        else return false;
        if ( last ) return true;
    }
}

uint32_t be32( const unsigned char* p ) {
    return ( (uint32_t)p[0] << 24 ) | ( (uint32_t)p[1] << 16 ) | ( (uint32_t)p[2] << 8 ) | p[3];
}

int paeth( int a, int b, int c ) {
    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return ( pa <= pb && pa <= pc ) ? a : ( pb <= pc ? b : c );
}

// Reverses the scanline filters of one (sub-)image of h lines of `line` bytes; in: h * (1 + line) bytes.
bool unfilter( const unsigned char* in, unsigned char* out, size_t h, size_t line, size_t bpp ) {
    const unsigned char* prev = nullptr;
    for ( size_t y = 0; y < h; y++ ) {
        const unsigned char f = in[y * ( line + 1 )];
        const unsigned char* s = in + y * ( line + 1 ) + 1;
        unsigned char* d       = out + y * line;
        for ( size_t i = 0; i < line; i++ ) {
            int a = i >= bpp ? d[i - bpp] : 0, b = prev ? prev[i] : 0, c = ( prev && i >= bpp ) ? prev[i - bpp] : 0;
            int v;
            switch ( f ) {
            case 0: v = s[i]; break;
            case 1: v = s[i] + a; break;
            case 2: v = s[i] + b; break;
            case 3: v = s[i] + ( ( a + b ) >> 1 ); break;
            case 4: v = s[i] + paeth( a, b, c ); break;
            default: return false;
            }
            d[i] = (unsigned char)v;
        }
        prev = d;
    }
    return true;
}

} // namespace

bool decode_png_rgb8( const unsigned char* file, size_t size, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgb ) {
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if ( size < 8 + 25 || std::memcmp( file, sig, 8 ) != 0 ) return false;
    size_t p = 8;
    int depth = 0, ctype = 0, interlace = 0;
    bool have_ihdr = false;
    std::vector<unsigned char> idat, palette;
    while ( p + 12 <= size ) {
        const uint32_t len        = be32( file + p );
        const unsigned char* type = file + p + 4;
        if ( len > size - p - 12 ) return false;
        const unsigned char* data = file + p + 8;
        if ( std::memcmp( type, "IHDR", 4 ) == 0 ) {
            if ( len != 13 ) return false;
            w = be32( data ), h = be32( data + 4 );
            depth = data[8], ctype = data[9], interlace = data[12];
            if ( data[10] != 0 || data[11] != 0 || interlace > 1 || w == 0 || h == 0 ) return false;
            have_ihdr = true;
        }
        else if ( std::memcmp( type, "PLTE", 4 ) == 0 ) palette.assign( data, data + len );
        else if ( std::memcmp( type, "IDAT", 4 ) == 0 ) idat.insert( idat.end(), data, data + len );
        else if ( std::memcmp( type, "IEND", 4 ) == 0 ) break;
        p += 12 + (size_t)len;
    }
    if ( !have_ihdr ) return false;
    int channels;
    switch ( ctype ) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return false;
    }
    const bool depth_ok = ( ctype == 0 ) ? ( depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16 )
                                         : ( ctype == 3 ? ( depth == 1 || depth == 2 || depth == 4 || depth == 8 ) : ( depth == 8 || depth == 16 ) );
    if ( !depth_ok ) return false;
    // a texture is at most 65535 x 65535 texels (the limit the device library applies as well)
    if ( w == 0 || h == 0 || w >= ( 1u << 16 ) || h >= ( 1u << 16 ) ) return false;
    const size_t bits_pp = (size_t)channels * depth, bpp = ( bits_pp + 7 ) / 8;

    // What the header promises, filter bytes included (an interlaced picture adds at most 15 bytes per
    // row of each of its seven sub-images).  Deflate expands at most 1032 : 1, so a header that promises
    // more than the file's compressed data can hold is wrong: nothing is allocated for it.
    const size_t expected = ( ( (size_t)w * bits_pp + 7 ) / 8 + 1 ) * h + ( interlace ? (size_t)7 * 16 * ( (size_t)h + 8 ) : 0 );
    if ( expected / 1032 > idat.size() + 1 ) return false;
    std::vector<unsigned char> raw;
    raw.reserve( expected );
    g_inflate_limit = expected;
    const bool inflated = zlib_inflate( idat.data(), idat.size(), raw );
    g_inflate_limit     = ~(size_t)0;
    if ( !inflated ) return false;

    rgb.assign( (size_t)w * h * 3, 0 );
    // one (sub-)image: its pixel (x, y) is pixel (x0 + x * dx, y0 + y * dy) of the picture
    auto emit = [&]( const unsigned char* px, size_t sw, size_t sh, size_t line, uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy ) -> bool {
        for ( size_t y = 0; y < sh; y++ ) {
            const unsigned char* row = px + y * line;
            for ( size_t x = 0; x < sw; x++ ) {
                unsigned char* o = &rgb[( (size_t)( y0 + y * dy ) * w + ( x0 + x * dx ) ) * 3];
                if ( depth == 8 || depth == 16 ) {
                    const unsigned char* s = row + x * bpp; // 16-bit samples: the high byte comes first
                    const size_t step      = depth / 8;
                    if ( ctype == 0 || ctype == 4 ) o[0] = o[1] = o[2] = s[0];
                    else if ( ctype == 3 ) {
                        if ( (size_t)s[0] * 3 + 2 >= palette.size() ) return false;
                        o[0] = palette[s[0] * 3], o[1] = palette[s[0] * 3 + 1], o[2] = palette[s[0] * 3 + 2];
                    }
                    else o[0] = s[0], o[1] = s[step], o[2] = s[2 * step];
                }
                else {
                    const size_t bit     = x * depth;
                    const unsigned int v = ( row[bit >> 3] >> ( 8 - depth - ( bit & 7 ) ) ) & ( ( 1u << depth ) - 1u );
                    if ( ctype == 3 ) {
                        if ( (size_t)v * 3 + 2 >= palette.size() ) return false;
                        o[0] = palette[v * 3], o[1] = palette[v * 3 + 1], o[2] = palette[v * 3 + 2];
                    }
                    else o[0] = o[1] = o[2] = (unsigned char)( ( v * 255u ) / ( ( 1u << depth ) - 1u ) );
                }
            }
        }
        return true;
    };

    if ( !interlace ) {
        const size_t line = ( (size_t)w * bits_pp + 7 ) / 8;
        if ( raw.size() < ( line + 1 ) * h ) return false;
        std::vector<unsigned char> px( line * h );
        if ( !unfilter( raw.data(), px.data(), h, line, bpp ) ) return false;
        return emit( px.data(), w, h, line, 0, 0, 1, 1 );
    }
    static const uint32_t ix[7] = { 0, 4, 0, 2, 0, 1, 0 }, iy[7] = { 0, 0, 4, 0, 2, 0, 1 }, sx[7] = { 8, 8, 4, 4, 2, 2, 1 }, sy[7] = { 8, 8, 8, 4, 4, 2, 2 };
    size_t at = 0;
    for ( int pass = 0; pass < 7; pass++ ) {
        const size_t sw = ( w + sx[pass] - ix[pass] - 1 ) / sx[pass], sh = ( h + sy[pass] - iy[pass] - 1 ) / sy[pass];
        if ( sw == 0 || sh == 0 ) continue;
        const size_t line = ( sw * bits_pp + 7 ) / 8;
        if ( raw.size() < at + ( line + 1 ) * sh ) return false;
        std::vector<unsigned char> px( line * sh );
        if ( !unfilter( raw.data() + at, px.data(), sh, line, bpp ) ) return false;
        if ( !emit( px.data(), sw, sh, line, ix[pass], iy[pass], sx[pass], sy[pass] ) ) return false;
        at += ( line + 1 ) * sh;
    }
    return true;
}

bool load_png_rgb8( const char* filename, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgb ) {
    FILE* f = std::fopen( filename, "rb" );
    if ( !f ) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t got;
    while ( ( got = std::fread( buf, 1, sizeof( buf ), f ) ) > 0 )
        file.insert( file.end(), buf, buf + got );
    std::fclose( f );
    return decode_png_rgb8( file.data(), file.size(), w, h, rgb );
}

} // namespace mrt
