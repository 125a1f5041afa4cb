// Host-side Scene / Transform / Camera / loaders of the GPU backend (see mrt_host.hpp).
#include "host_math.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <sstream>
#include <stdexcept>

namespace mrt {

static const double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------ Transform (defines.h:67-148)
vec3 Transform::transformTo( vec3 const& p ) const {
    return mul( m_invTransform, p - m_translation );
}
vec3 Transform::transformFrom( vec3 const& p ) const {
    return mul( m_transform, p ) + m_translation;
}
vec3 Transform::vectorTransformFrom( vec3 const& dir ) const {
    const mat3& m = m_invTransform;
    return vec3( dot( vec3( m.c[0][0], m.c[0][1], m.c[0][2] ), dir ),
                 dot( vec3( m.c[1][0], m.c[1][1], m.c[1][2] ), dir ),
                 dot( vec3( m.c[2][0], m.c[2][1], m.c[2][2] ), dir ) );
}
void Transform::translate( vec3 const& p ) {
    m_translation = m_translation + p;
}
void Transform::transform( mat3 const& m ) {
    m_transform    = mul( m, m_transform );
    m_translation  = mul( m, m_translation );
    m_invTransform = inverse( m_transform );
}
void Transform::scale( float sx, float sy, float sz ) {
    mat3 m    = zero3();
    m.c[0][0] = sx;
    m.c[1][1] = sy;
    m.c[2][2] = sz;
    transform( m );
}
void Transform::rotate( vec3 const& axis, float degrees ) {
    // degrees_to_radians (defines.h:53-55) evaluates in double and returns a float
    float theta = (float)( (double)degrees * kPi / (double)180.0f );
    float c     = std::cos( theta );
    if ( c == 1 ) return;
    float s   = std::sin( theta );
    float t   = 1 - c;
    float tx  = t * axis.x;
    float ty  = t * axis.y;
    float tz  = t * axis.z;
    float txy = tx * axis.y;
    float txz = tx * axis.z;
    float tyz = ty * axis.z;
    float sx  = s * axis.x;
    float sy  = s * axis.y;
    float sz  = s * axis.z;
    mat3 m    = zero3();
    m.c[0][0] = tx * axis.x + c;
    m.c[0][1] = txy + sz;
    m.c[0][2] = txz - sy;
    m.c[1][0] = txy - sz;
    m.c[1][1] = ty * axis.y + c;
    m.c[1][2] = tyz + sx;
    m.c[2][0] = txz + sy;
    m.c[2][1] = tyz - sx;
    m.c[2][2] = tz * axis.z + c;
    transform( m );
}
bool Transform::operator==( Transform const& o ) const {
    return std::memcmp( &m_transform, &o.m_transform, sizeof( mat3 ) ) == 0 &&
           std::memcmp( &m_invTransform, &o.m_invTransform, sizeof( mat3 ) ) == 0 &&
           std::memcmp( &m_translation, &o.m_translation, sizeof( vec3 ) ) == 0;
}
bool Transform::isIdentity() const {
    return *this == Transform();
}

// ------------------------------------------------------------------ Material (Material.cpp:241-262)
Material::Material( size_t UID, MaterialModel matModel, MatType matType ) : m_MatModel( matModel ), m_UID( UID ) {
    if ( matModel == MaterialModel::COOK_TORRANCE ) {
        m_IOR       = 1.0;
        m_roughness = 0.1;
        m_metalness = 0.001f;
        m_albedo    = color3( 1.f );
        m_MatType   = matType;
    }
    else {
        m_IOR    = 1.0;
        m_albedo = color3( 0.5f );
    }
}

// ------------------------------------------------------------------ scene.cpp:33-175
Scene::~Scene() {
    for ( auto* o : objects )
        delete o;
    for ( auto* l : lights )
        delete l;
    for ( auto* t : owned_textures )
        delete t;
    delete cam;
}
Scene* initScene() {
    return new Scene;
}
void freeScene( Scene* scene ) {
    delete scene;
}
Object* initSphere( int materialIndex, Transform transform ) {
    auto ret                = new Object( materialIndex, transform );
    ret->geom.type          = SPHERE;
    ret->geom.sphere.center = transform.transformFrom( vec3( 0.f ) );
    const mat3& m           = transform.getTransform();
    float s0 = dot( vec3( m.c[0][0], m.c[0][1], m.c[0][2] ), vec3( m.c[0][0], m.c[0][1], m.c[0][2] ) );
    float s1 = dot( vec3( m.c[1][0], m.c[1][1], m.c[1][2] ), vec3( m.c[1][0], m.c[1][1], m.c[1][2] ) );
    float s2 = dot( vec3( m.c[2][0], m.c[2][1], m.c[2][2] ), vec3( m.c[2][0], m.c[2][1], m.c[2][2] ) );
    ret->geom.sphere.radius = 1.f * std::sqrt( std::max( s0, std::max( s1, s2 ) ) );
    return ret;
}
Object* initSmoothTriangle( vec3 p1, vec3 p2, vec3 p3, vec3 n, vec2 t[3], vec3 n1, vec3 n2, vec3 n3, Transform transform, int materialIndex ) {
    auto ret                  = new Object( materialIndex, transform );
    ret->geom.type            = TRIANGLE;
    ret->geom.triangle.p1     = p1;
    ret->geom.triangle.p2     = p2;
    ret->geom.triangle.p3     = p3;
    ret->geom.triangle.n1     = n1;
    ret->geom.triangle.n2     = n2;
    ret->geom.triangle.n3     = n3;
    ret->geom.triangle.normal = n;
    ret->geom.triangle.tex[0] = t[0];
    ret->geom.triangle.tex[1] = t[1];
    ret->geom.triangle.tex[2] = t[2];
    return ret;
}
Object* initTriangle( vec3 p1, vec3 p2, vec3 p3, vec3 n, vec2 t[3], Transform transform, int materialIndex ) {
    // The reference leaves n1..n3 unset here (scene.cpp:65-83) and then interpolates them
    // (triangle.cpp:41); the face normal is the only defined choice.
    return initSmoothTriangle( p1, p2, p3, n, t, n, n, n, transform, materialIndex );
}
void addObject( Scene* scene, Object* obj ) {
    scene->objects.push_back( obj );
}
void addLight( Scene* scene, Light* light ) {
    scene->lights.push_back( light );
}
void setSimpleCamera( Scene* scene, point3 position, vec3 at, vec3 up, float fov, float w, float h, float radius, float focusDist ) {
    delete scene->cam;
    scene->cam = new SimpleCamera( position, at, up, fov, ( w / h ), (int)w, (int)h, radius, focusDist );
}

// ------------------------------------------------------------------ SimpleCamera (Camera.cpp:5-35)
SimpleCamera::SimpleCamera( point3 lookfrom, point3 lookat, vec3 vup, float vfov, float aspect_ratio, int width, int height, float lensRadius, float focalDistance ) {
    pos             = lookfrom;
    fov             = vfov;
    aspect          = aspect_ratio;
    m_lensRadius    = lensRadius;
    m_focalDistance = focalDistance;
    imgWidth        = (float)width;
    imgHeight       = (float)height;
    dir             = lookat;
    dir             = dir - pos;
    dir             = normalize( dir );
    xDir            = cross( dir, vup );
    up              = normalize( cross( xDir, dir ) );
    left            = normalize( cross( up, dir ) );
    float radfov    = (float)( (double)( fov * (float)kPi ) / 180.0 );
    l               = m_focalDistance;
    h               = std::tan( radfov * .5f ) * ( 2.0f * l );
    w               = h * imgWidth / (float)imgHeight;
    point3 B         = pos + ( l * dir ) + ( h / 2.0f * up );
    nearPlaneTopLeft = B + w / 2.0f * ( left );
    dXPixel          = -left * ( w / (float)imgWidth );
    dYPixel          = -up * ( w / (float)imgWidth );
}

// ------------------------------------------------------------------ textures
texture* checker_texture( color3 c1, color3 c2 ) {
    auto* t = new texture;
    t->kind = MRT_TEX_CHECKER;
    t->even = c1;
    t->odd  = c2;
    return t;
}
// image_texture( filename ) (textures.hpp:50-62 -> loadPng, image.cpp:13-40): a PNG file, or a
// binary PPM (P6, maxval 255) holding the same bytes.  The value mapping is loadPng's, byte / 255.f.
static texture* image_texture_unguarded( const char* filename );
texture* image_texture( const char* filename ) {
    try { // a file that cannot be read for whatever reason, memory included, is "no texture", not an exception
        return image_texture_unguarded( filename );
    }
    catch ( const std::exception& ) {
        return nullptr;
    }
}
static texture* image_texture_unguarded( const char* filename ) {
    FILE* f = std::fopen( filename, "rb" );
    if ( !f ) return nullptr;
    unsigned char sig[2] = { 0, 0 };
    size_t n_sig         = std::fread( sig, 1, 2, f );
    std::rewind( f );
    uint32_t w = 0, h = 0;
    std::vector<unsigned char> buf;
    if ( n_sig == 2 && sig[0] == 0x89 && sig[1] == 'P' ) {
        std::fclose( f );
        if ( !load_png_rgb8( filename, w, h, buf ) ) return nullptr;
    }
    else {
        char magic[3] = { 0 };
        int pw = 0, ph = 0, maxv = 0;
        if ( std::fscanf( f, "%2s %d %d %d", magic, &pw, &ph, &maxv ) != 4 || std::strcmp( magic, "P6" ) != 0 || maxv != 255 || pw <= 0 || ph <= 0 ) {
            std::fclose( f );
            return nullptr;
        }
        std::fgetc( f );
        w = (uint32_t)pw, h = (uint32_t)ph;
        // the header must not promise more than the file holds (nor more than a texture may have)
        const long at = std::ftell( f );
        std::fseek( f, 0, SEEK_END );
        const long file_end = std::ftell( f );
        if ( at < 0 || file_end < at || w >= ( 1u << 16 ) || h >= ( 1u << 16 ) || (uint64_t)w * h * 3 > (uint64_t)( file_end - at ) ) {
            std::fclose( f );
            return nullptr;
        }
        std::fseek( f, at, SEEK_SET );
        buf.resize( (size_t)w * h * 3 );
        size_t got = std::fread( buf.data(), 1, buf.size(), f );
        std::fclose( f );
        if ( got != buf.size() ) return nullptr;
    }
    auto* t   = new texture;
    t->kind   = MRT_TEX_IMAGE;
    t->width  = w;
    t->height = h;
    t->texels.resize( buf.size() );
    for ( size_t i = 0; i < buf.size(); i++ )
        t->texels[i] = (float)(unsigned int)buf[i] / 255.f;
    return t;
}

// ------------------------------------------------------------------ IBL: Radiance RGBE reader
// Pixel decode is the one stb_image's stbi_loadf applies to .hdr files, which the reference calls
// (Light.h:205): value = mantissa * 2^(exponent - 136), exponent 0 -> 0; new-style RLE scanlines.
static void rgbe_to_float( const unsigned char* in, float* out ) {
    if ( in[3] != 0 ) {
        float f1 = std::ldexp( 1.0f, (int)in[3] - ( 128 + 8 ) );
        out[0]   = in[0] * f1;
        out[1]   = in[1] * f1;
        out[2]   = in[2] * f1;
    }
    else { out[0] = out[1] = out[2] = 0.f; }
}
IBL::IBL( int width, int height, std::vector<float> rgb ) : m_width( width ), m_height( height ), m_pixels( std::move( rgb ) ) {
    m_color = color3( 0.f );
}
IBL::IBL( const std::string& filename ) {
    m_color = color3( 0.f );
    std::ifstream f( filename, std::ios::binary );
    if ( !f ) throw std::runtime_error( "IBL: cannot open " + filename );
    std::string line;
    std::getline( f, line );
    if ( line.rfind( "#?RADIANCE", 0 ) != 0 && line.rfind( "#?RGBE", 0 ) != 0 ) throw std::runtime_error( "IBL: not a Radiance file: " + filename );
    bool fmt = false;
    while ( std::getline( f, line ) ) {
        if ( line.empty() ) break;
        if ( line == "FORMAT=32-bit_rle_rgbe" ) fmt = true;
    }
    if ( !fmt ) throw std::runtime_error( "IBL: unsupported format in " + filename );
    std::getline( f, line );
    int w = 0, h = 0;
    if ( std::sscanf( line.c_str(), "-Y %d +X %d", &h, &w ) != 2 ) throw std::runtime_error( "IBL: unsupported orientation in " + filename );
    m_width  = w;
    m_height = h;
    m_pixels.resize( (size_t)w * h * 3 );
    std::vector<unsigned char> rest( ( std::istreambuf_iterator<char>( f ) ), std::istreambuf_iterator<char>() );
    size_t p = 0;
    auto need = [&]( size_t n ) {
        if ( p + n > rest.size() ) throw std::runtime_error( "IBL: truncated file " + filename );
    };
    std::vector<unsigned char> scan( (size_t)w * 4 );
    if ( w < 8 || w >= 32768 ) { // flat data
        for ( size_t i = 0; i < (size_t)w * h; i++ ) {
            need( 4 );
            rgbe_to_float( &rest[p], &m_pixels[3 * i] );
            p += 4;
        }
        return;
    }
    for ( int j = 0; j < h; j++ ) {
        need( 4 );
        int c1 = rest[p], c2 = rest[p + 1], len = rest[p + 2];
        if ( c1 != 2 || c2 != 2 || ( len & 0x80 ) ) { // not run-length encoded: flat from here
            if ( j != 0 ) throw std::runtime_error( "IBL: mixed encoding in " + filename );
            for ( size_t i = 0; i < (size_t)w * h; i++ ) {
                need( 4 );
                rgbe_to_float( &rest[p], &m_pixels[3 * i] );
                p += 4;
            }
            return;
        }
        len = ( len << 8 ) | rest[p + 3];
        p += 4;
        if ( len != w ) throw std::runtime_error( "IBL: bad scanline width in " + filename );
        for ( int k = 0; k < 4; k++ ) {
            int i = 0;
            while ( i < w ) {
                need( 1 );
                int count = rest[p++];
                if ( count > 128 ) {
                    count -= 128;
                    need( 1 );
                    unsigned char value = rest[p++];
                    if ( i + count > w ) throw std::runtime_error( "IBL: corrupt run in " + filename );
                    for ( int z = 0; z < count; z++ )
                        scan[( i++ ) * 4 + k] = value;
                }
                else {
                    need( count );
                    if ( count == 0 || i + count > w ) throw std::runtime_error( "IBL: corrupt run in " + filename );
                    for ( int z = 0; z < count; z++ )
                        scan[( i++ ) * 4 + k] = rest[p++];
                }
            }
        }
        for ( int i = 0; i < w; i++ )
            rgbe_to_float( &scan[i * 4], &m_pixels[( (size_t)j * w + i ) * 3] );
    }
}

// ------------------------------------------------------------------ OBJ / MTL
namespace {

struct MtlRec {
    float Kd[3] = { 0.f, 0.f, 0.f }, Ke[3] = { 0, 0, 0 }, Ks[3] = { 0, 0, 0 }; // InitMaterial (tiny_obj_loader.h:853-888)
    float Ni    = 1.f;
    std::string name;
};

struct ObjData {
    std::vector<float> v, vn, vt;
    struct Corner {
        int v, vt, vn;
    };
    std::vector<Corner> corners;  // 3 per triangle (polygons are fanned like tinyobj's triangulate)
    std::vector<int> face_mtl;    // per triangle, -1 = none
    std::vector<MtlRec> mtls;
};

// What follows restates the OBJ / MTL reading of the reference's tiny_obj_loader.h (v1.0.6) statement
// by statement: which faces exist, their order (it numbers the primitives) and every float depend on
// its grammar and on its arithmetic, neither of which is strtod's.  tiny_obj_loader.h is third-party
// code the reference vendors under the MIT licence; the number parser below (parse_double, parse_triple,
// fix_index) follows its tryParseDouble / parseReal / fixIndex closely enough to carry its notice:
//
//   The MIT License (MIT)
//   Copyright (c) 2012-2016 Syoyo Fujita and many contributors.
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and
//   associated documentation files (the "Software"), to deal in the Software without restriction,
//   including without limitation the rights to use, copy, modify, merge, publish, distribute,
//   sublicense, and/or sell copies of the Software, and to permit persons to whom the Software is
//   furnished to do so, subject to the following conditions:
//   The above copyright notice and this permission notice shall be included in all copies or
//   substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT
//   NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
//   NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM,
//   DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT
//   OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.

inline bool is_space( char c ) {
    return c == ' ' || c == '\t';
}
inline bool is_digit( char c ) {
    return (unsigned)( c - '0' ) < 10u;
}
inline bool is_new_line( char c ) {
    return c == '\r' || c == '\n' || c == '\0';
}

// The lines of a file as tinyobj's reader cuts them (safeGetline, tiny_obj_loader.h:389-417): a line
// ends at \n, \r\n or a lone \r; a last line without an end still counts.  A stream that opens but
// cannot be read (a directory) has no lines.
std::vector<std::string> read_lines( std::istream& is ) {
    std::string data;
    try {
        data.assign( ( std::istreambuf_iterator<char>( is ) ), std::istreambuf_iterator<char>() );
    }
    catch ( const std::exception& ) { // what the stream buffer of a directory throws; the reference's peek() turns it into "no more lines"
        data.clear();
    }
    std::vector<std::string> lines;
    size_t from = 0;
    for ( size_t i = 0; i < data.size(); i++ ) {
        if ( data[i] != '\n' && data[i] != '\r' ) continue;
        lines.emplace_back( data, from, i - from );
        if ( data[i] == '\r' && i + 1 < data.size() && data[i + 1] == '\n' ) i++;
        from = i + 1;
    }
    if ( from < data.size() ) lines.emplace_back( data, from, data.size() - from );
    return lines;
}

// tryParseDouble (tiny_obj_loader.h:474-587).  Grammar: [sign] digits ["." digits] [("e"|"E") [sign]
// digits]; a number has to start with a sign or a digit and have a digit before any "." (".5", "-.5"
// and "-" fail; "5." is 5).  A word that fails reads as the default, 0.  Arithmetic: the integer digits accumulate as m = m * 10 + d, the
// k-th decimal adds d * 10^-k (a table of literals for k < 8, pow(10, -k) beyond), an exponent e
// applies as ldexp(m * pow(5, e), e): a few units in the last place of the double away from the
// correctly rounded value, which shows in the float now and then.
bool parse_double( const char* s, const char* s_end, double* result ) {
    if ( s >= s_end ) return false;
    double mantissa = 0.0;
    int exponent    = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read         = 0;
    bool more        = false; // curr has not reached s_end
    if ( *curr == '+' || *curr == '-' ) {
        sign = *curr;
        curr++;
    }
    else if ( !is_digit( *curr ) ) return false;
    more = curr != s_end;
    while ( more && is_digit( *curr ) ) {
        mantissa *= 10;
        mantissa += (int)( *curr - '0' );
        curr++, read++;
        more = curr != s_end;
    }
    if ( read == 0 ) return false;
    bool assemble = !more;
    if ( !assemble ) {
        if ( *curr == '.' ) {
            curr++;
            read = 1;
            more = curr != s_end;
            while ( more && is_digit( *curr ) ) {
                static const double pow_lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
                const int lut_entries         = 8;
                mantissa += (int)( *curr - '0' ) * ( read < lut_entries ? pow_lut[read] : std::pow( 10.0, -read ) );
                read++, curr++;
                more = curr != s_end;
            }
        }
        else if ( *curr != 'e' && *curr != 'E' ) { assemble = true; }
        if ( !assemble && more && ( *curr == 'e' || *curr == 'E' ) ) {
            curr++;
            more = curr != s_end;
            // at s_end the reference looks at the character there (a blank, a \r or the string's end): no digit
            if ( more && ( *curr == '+' || *curr == '-' ) ) {
                exp_sign = *curr;
                curr++;
            }
            else if ( !( more && is_digit( *curr ) ) ) return false;
            read = 0;
            more = curr != s_end;
            while ( more && is_digit( *curr ) ) {
                exponent *= 10;
                exponent += (int)( *curr - '0' );
                curr++, read++;
                more = curr != s_end;
            }
            exponent *= ( exp_sign == '+' ? 1 : -1 );
            if ( read == 0 ) return false;
        }
    }
    *result = ( sign == '+' ? 1 : -1 ) * ( exponent ? std::ldexp( mantissa * std::pow( 5.0, exponent ), exponent ) : mantissa );
    return true;
}

// parseReal (tiny_obj_loader.h:589-597): the next blank-delimited word, default_value when it is no number
float parse_real( const char** token, double default_value = 0.0 ) {
    ( *token ) += std::strspn( *token, " \t" );
    const char* end = ( *token ) + std::strcspn( *token, " \t\r" );
    double val      = default_value;
    parse_double( *token, end, &val );
    *token = end;
    return (float)val;
}

// sscanf( token, "%s", namebuf ): the first word after any white space (empty when there is none)
std::string scan_word( const char* token ) {
    while ( *token && std::isspace( (unsigned char)*token ) )
        token++;
    const char* e = token;
    while ( *e && !std::isspace( (unsigned char)*e ) )
        e++;
    return std::string( token, e );
}

// LoadMtl (tiny_obj_loader.h:965-1333) for the four statements the scene builder reads.  A material
// is stored when the next newmtl arrives (if it has a name) and the last one always, named or not;
// of two materials of one name the first is the one usemtl finds.
void parse_mtl( std::istream& f, std::vector<MtlRec>& out, std::map<std::string, int>& names ) {
    MtlRec material;
    for ( std::string& line : read_lines( f ) ) {
        if ( !line.empty() ) line = line.substr( 0, line.find_last_not_of( " \t" ) + 1 ); // trailing blanks are cut (MTL only)
        if ( line.empty() ) continue;
        const char* token = line.c_str();
        token += std::strspn( token, " \t" );
        if ( token[0] == '\0' || token[0] == '#' ) continue;
        if ( std::strncmp( token, "newmtl", 6 ) == 0 && is_space( token[6] ) ) {
            if ( !material.name.empty() ) {
                names.insert( { material.name, (int)out.size() } );
                out.push_back( material );
            }
            material      = MtlRec();
            material.name = scan_word( token + 7 );
            continue;
        }
        auto rd3 = [&]( float* d ) {
            token += 2;
            d[0] = parse_real( &token );
            d[1] = parse_real( &token );
            d[2] = parse_real( &token );
        };
        if ( token[0] == 'K' && token[1] == 'd' && is_space( token[2] ) ) rd3( material.Kd );
        else if ( token[0] == 'K' && token[1] == 's' && is_space( token[2] ) ) rd3( material.Ks );
        else if ( token[0] == 'K' && token[1] == 'e' && is_space( token[2] ) ) rd3( material.Ke );
        else if ( token[0] == 'N' && token[1] == 'i' && is_space( token[2] ) ) {
            token += 2;
            material.Ni = parse_real( &token );
        }
    }
    names.insert( { material.name, (int)out.size() } );
    out.push_back( material );
}

int fix_index( int idx, int n ) { // fixIndex (tiny_obj_loader.h:425-429): 1-based, negative = relative to the end, 0 stays 0
    if ( idx > 0 ) return idx - 1;
    if ( idx == 0 ) return 0;
    return n + idx;
}

// parseTriple (tiny_obj_loader.h:691-722): i, i/j, i//k, i/j/k
ObjData::Corner parse_triple( const char** token, int vsize, int vnsize, int vtsize ) {
    ObjData::Corner vi { -1, -1, -1 };
    vi.v = fix_index( std::atoi( *token ), vsize );
    ( *token ) += std::strcspn( *token, "/ \t\r" );
    if ( ( *token )[0] != '/' ) return vi;
    ( *token )++;
    if ( ( *token )[0] == '/' ) {
        ( *token )++;
        vi.vn = fix_index( std::atoi( *token ), vnsize );
        ( *token ) += std::strcspn( *token, "/ \t\r" );
        return vi;
    }
    vi.vt = fix_index( std::atoi( *token ), vtsize );
    ( *token ) += std::strcspn( *token, "/ \t\r" );
    if ( ( *token )[0] != '/' ) return vi;
    ( *token )++;
    vi.vn = fix_index( std::atoi( *token ), vnsize );
    ( *token ) += std::strcspn( *token, "/ \t\r" );
    return vi;
}

// LoadObj (tiny_obj_loader.h:1424-1730) with triangulation.  Faces collect in a group; usemtl with
// another material moves the group into the current shape, g / o close the shape.  One behaviour of
// that version is kept because it decides which primitives exist: when g or o follows a usemtl
// with no face in between, the faces the shape already holds are dropped with it (the shape is
// only kept when the group that closes it is not empty); at the end of the file they are kept.
void parse_obj( const char* filename, ObjData& d ) {
    std::ifstream f( filename );
    if ( !f ) throw std::runtime_error( std::string( "cannot open OBJ file " ) + filename );
    std::string dir = filename;
    size_t slash    = dir.find_last_of( "/\\" );
    dir             = ( slash == std::string::npos ) ? std::string( "" ) : dir.substr( 0, slash + 1 );
    std::map<std::string, int> mtl_names;
    int material = -1;
    std::vector<std::vector<ObjData::Corner>> group; // faceGroup
    std::vector<ObjData::Corner> shape_corners;       // shape.mesh.indices
    std::vector<int> shape_mtl;                       // shape.mesh.material_ids
    auto export_group = [&]() {                       // exportFaceGroupToShape (tiny_obj_loader.h:890-951)
        if ( group.empty() ) return false;
        for ( const auto& face : group )
            for ( size_t k = 2; k < face.size(); k++ ) {
                shape_corners.push_back( face[0] );
                shape_corners.push_back( face[k - 1] );
                shape_corners.push_back( face[k] );
                shape_mtl.push_back( material );
            }
        return true;
    };
    auto close_shape = [&]( bool keep ) {
        if ( keep ) {
            d.corners.insert( d.corners.end(), shape_corners.begin(), shape_corners.end() );
            d.face_mtl.insert( d.face_mtl.end(), shape_mtl.begin(), shape_mtl.end() );
        }
        shape_corners.clear();
        shape_mtl.clear();
        group.clear();
    };
    for ( const std::string& line : read_lines( f ) ) {
        if ( line.empty() ) continue;
        const char* token = line.c_str();
        token += std::strspn( token, " \t" );
        if ( token[0] == '\0' || token[0] == '#' ) continue;
        if ( token[0] == 'v' && is_space( token[1] ) ) {
            token += 2;
            for ( int k = 0; k < 3; k++ )
                d.v.push_back( parse_real( &token ) );
        }
        else if ( token[0] == 'v' && token[1] == 'n' && is_space( token[2] ) ) {
            token += 3;
            for ( int k = 0; k < 3; k++ )
                d.vn.push_back( parse_real( &token ) );
        }
        else if ( token[0] == 'v' && token[1] == 't' && is_space( token[2] ) ) {
            token += 3;
            for ( int k = 0; k < 2; k++ )
                d.vt.push_back( parse_real( &token ) );
        }
        else if ( token[0] == 'f' && is_space( token[1] ) ) {
            token += 2;
            token += std::strspn( token, " \t" );
            std::vector<ObjData::Corner> face;
            while ( !is_new_line( token[0] ) ) {
                face.push_back( parse_triple( &token, (int)( d.v.size() / 3 ), (int)( d.vn.size() / 3 ), (int)( d.vt.size() / 2 ) ) );
                token += std::strspn( token, " \t\r" );
            }
            group.push_back( std::move( face ) );
        }
        else if ( std::strncmp( token, "usemtl", 6 ) == 0 && is_space( token[6] ) ) {
            auto it            = mtl_names.find( scan_word( token + 7 ) );
            const int new_mtl  = ( it == mtl_names.end() ) ? -1 : it->second;
            if ( new_mtl != material ) {
                export_group();
                group.clear();
                material = new_mtl;
            }
        }
        else if ( std::strncmp( token, "mtllib", 6 ) == 0 && is_space( token[6] ) ) {
            // the names are split at single blanks and tried in turn until one file opens (an empty
            // name, as two blanks in a row make one, opens the directory and yields one unnamed
            // default material, exactly as the stream the reference opens on it does)
            std::stringstream ss( std::string( token + 7 ) );
            std::string name;
            while ( std::getline( ss, name, ' ' ) ) {
                std::ifstream mf( ( dir + name ).c_str() );
                if ( !mf ) continue;
                parse_mtl( mf, d.mtls, mtl_names );
                break;
            }
        }
        else if ( ( token[0] == 'g' || token[0] == 'o' ) && is_space( token[1] ) ) {
            close_shape( export_group() );
        }
    }
    const bool ret = export_group();
    close_shape( ret || !shape_corners.empty() );
}

struct Tri {
    vec3 p[3], n[3];
    vec2 t[3];
    bool has_n;
    int mtl;
};

void subdivide_once( std::vector<Tri>& tris ) {
    std::vector<Tri> out;
    out.reserve( tris.size() * 4 );
    auto mid3 = []( vec3 a, vec3 b ) { return ( a + b ) * 0.5f; };
    auto mid2 = []( vec2 a, vec2 b ) { return vec2 { ( a.x + b.x ) * 0.5f, ( a.y + b.y ) * 0.5f }; };
    for ( const Tri& t : tris ) {
        vec3 m01 = mid3( t.p[0], t.p[1] ), m12 = mid3( t.p[1], t.p[2] ), m20 = mid3( t.p[2], t.p[0] );
        vec3 n01 = mid3( t.n[0], t.n[1] ), n12 = mid3( t.n[1], t.n[2] ), n20 = mid3( t.n[2], t.n[0] );
        vec2 t01 = mid2( t.t[0], t.t[1] ), t12 = mid2( t.t[1], t.t[2] ), t20 = mid2( t.t[2], t.t[0] );
        Tri a    = t, b = t, c = t, m = t;
        a.p[1] = m01, a.p[2] = m20, a.n[1] = n01, a.n[2] = n20, a.t[1] = t01, a.t[2] = t20;
        b.p[0] = m01, b.p[2] = m12, b.n[0] = n01, b.n[2] = n12, b.t[0] = t01, b.t[2] = t12;
        c.p[0] = m20, c.p[1] = m12, c.n[0] = n20, c.n[1] = n12, c.t[0] = t20, c.t[1] = t12;
        m.p[0] = m01, m.p[1] = m12, m.p[2] = m20, m.n[0] = n01, m.n[1] = n12, m.n[2] = n20, m.t[0] = t01, m.t[1] = t12, m.t[2] = t20;
        out.push_back( a );
        out.push_back( b );
        out.push_back( c );
        out.push_back( m );
    }
    tris.swap( out );
}

} // namespace

void addObjectsFromFileEx( const char* filename, Scene* scene, size_t default_mat, Transform& transform, float prescale, int subdivisions ) {
    ObjData d;
    parse_obj( filename, d );
    std::vector<Tri> tris;
    tris.reserve( d.face_mtl.size() );
    for ( size_t f = 0; f < d.face_mtl.size(); f++ ) {
        Tri t {};
        t.has_n = !d.vn.empty();
        t.mtl   = d.face_mtl[f];
        for ( int k = 0; k < 3; k++ ) {
            const ObjData::Corner& c = d.corners[3 * f + k];
            // the reference reads whatever lies at an index outside the file's tables; refused here
            if ( c.v < 0 || (size_t)c.v >= d.v.size() / 3 || ( t.has_n && c.vn >= 0 && (size_t)c.vn >= d.vn.size() / 3 ) ||
                 ( !d.vt.empty() && c.vt >= 0 && (size_t)c.vt >= d.vt.size() / 2 ) )
                throw std::runtime_error( std::string( "OBJ file with a face index outside its vertex tables: " ) + filename );
            t.p[k] = vec3( d.v[3 * c.v] * prescale, d.v[3 * c.v + 1] * prescale, d.v[3 * c.v + 2] * prescale );
            if ( t.has_n && c.vn >= 0 ) t.n[k] = vec3( d.vn[3 * c.vn], d.vn[3 * c.vn + 1], d.vn[3 * c.vn + 2] );
            if ( !d.vt.empty() && c.vt >= 0 ) t.t[k] = vec2 { d.vt[2 * c.vt], d.vt[2 * c.vt + 1] };
        }
        tris.push_back( t );
    }
    for ( int s = 0; s < subdivisions; s++ )
        subdivide_once( tris );

    // example_scenes.cpp:34-151
    for ( const Tri& t : tris ) {
        vec3 v0 = t.p[0], v1 = t.p[1], v2 = t.p[2];
        vec3 normals[3] = { t.n[0], t.n[1], t.n[2] };
        vec2 textures[3] = { t.t[0], t.t[1], t.t[2] };
        vec3 n;
        if ( !t.has_n ) {
            vec3 v2v0  = v2 - v0;
            vec3 v1v0  = v1 - v0;
            n          = normalize( cross( v1v0, v2v0 ) );
            normals[0] = normals[1] = normals[2] = n;
        }
        else { n = normalize( normals[0] + normals[1] + normals[2] ); }

        if ( !d.mtls.empty() ) {
            const MtlRec& mr = d.mtls[t.mtl >= 0 ? t.mtl : 0];
            auto& mat        = scene->CreateMaterial( MaterialModel::COOK_TORRANCE );
            mat.m_IOR        = mr.Ni;
            mat.m_roughness  = 1.f;
            mat.m_metalness  = 0.0;
            mat.m_albedo     = color3( mr.Kd[0], mr.Kd[1], mr.Kd[2] );
            if ( mr.Ke[0] != 0 || mr.Ke[1] != 0 || mr.Ke[2] != 0 ) {
                mat.m_emission = color3( mr.Ke[0], mr.Ke[1], mr.Ke[2] );
                auto lightObj  = initSmoothTriangle( v0, v1, v2, n, textures, normals[0], normals[1], normals[2], transform, mat.m_UID );
                addLight( scene, new ShapeLight( vec3( 0.f ), mat.m_emission, lightObj ) );
            }
            if ( mr.Ks[0] == 1 && mr.Ks[1] == 1 && mr.Ks[2] == 1 ) {
                mat.m_MatType   = SPECULAR;
                mat.m_metalness = 1.0;
                mat.m_roughness = 0.002;
            }
            addObject( scene, initSmoothTriangle( v0, v1, v2, n, textures, normals[0], normals[1], normals[2], transform, mat.m_UID ) );
        }
        else {
            addObject( scene, initSmoothTriangle( v0, v1, v2, n, textures, normals[0], normals[1], normals[2], transform, (int)default_mat ) );
        }
    }
}

void addObjectsFromFile( const char* filename, Scene* scene, Material& default_mat, Transform& transform ) {
    addObjectsFromFileEx( filename, scene, default_mat.m_UID, transform, 1.f, 0 );
}

bool savePPM( const char* filename, const float* rgb, uint32_t w, uint32_t h ) {
    FILE* f = std::fopen( filename, "wb" );
    if ( !f ) return false;
    std::fprintf( f, "P6\n%u %u\n255\n", w, h );
    std::vector<unsigned char> row( (size_t)w * 3 );
    for ( uint32_t y = 0; y < h; y++ ) {
        for ( uint32_t i = 0; i < w * 3; i++ ) {
            int c  = (int)( 255.f * rgb[(size_t)y * w * 3 + i] ); // ivec3(255.f * c), clamp 0..255
            row[i] = (unsigned char)( c < 0 ? 0 : ( c > 255 ? 255 : c ) );
        }
        std::fwrite( row.data(), 1, row.size(), f );
    }
    std::fclose( f );
    return true;
}

// ------------------------------------------------------------------ PNG (image.cpp:127-146)
// saveImage converts every channel with ivec3(255.f * c) clamped to 0..255 and hands the bytes to
// lodepng_encode24_file.  Same bytes here; the container is a minimal PNG: 8-bit RGB, no filtering,
// the zlib stream made of stored (uncompressed) deflate blocks.
namespace {
uint32_t crc32_update( uint32_t crc, const unsigned char* p, size_t n ) {
    static uint32_t table[256];
    static bool init = false;
    if ( !init ) {
        for ( uint32_t i = 0; i < 256; i++ ) {
            uint32_t c = i;
            for ( int k = 0; k < 8; k++ )
                c = ( c & 1u ) ? 0xedb88320u ^ ( c >> 1 ) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for ( size_t i = 0; i < n; i++ )
        crc = table[( crc ^ p[i] ) & 0xffu] ^ ( crc >> 8 );
    return crc;
}
void put_be32( std::vector<unsigned char>& v, uint32_t x ) {
    v.push_back( (unsigned char)( x >> 24 ) ), v.push_back( (unsigned char)( x >> 16 ) ), v.push_back( (unsigned char)( x >> 8 ) ), v.push_back( (unsigned char)x );
}
bool write_chunk( FILE* f, const char* type, const std::vector<unsigned char>& data ) {
    std::vector<unsigned char> head;
    put_be32( head, (uint32_t)data.size() );
    uint32_t crc = crc32_update( 0xffffffffu, (const unsigned char*)type, 4 );
    if ( !data.empty() ) crc = crc32_update( crc, data.data(), data.size() );
    std::vector<unsigned char> tail;
    put_be32( tail, crc ^ 0xffffffffu );
    return std::fwrite( head.data(), 1, 4, f ) == 4 && std::fwrite( type, 1, 4, f ) == 4 &&
           ( data.empty() || std::fwrite( data.data(), 1, data.size(), f ) == data.size() ) && std::fwrite( tail.data(), 1, 4, f ) == 4;
}
} // namespace

bool savePNG( const char* filename, const float* rgb, uint32_t w, uint32_t h ) {
    if ( w == 0 || h == 0 ) return false;
    // scanlines: filter byte 0 + w * 3 bytes
    const size_t row = (size_t)w * 3 + 1;
    std::vector<unsigned char> raw( row * h );
    for ( uint32_t y = 0; y < h; y++ ) {
        raw[row * y] = 0;
        for ( uint32_t i = 0; i < w * 3; i++ ) {
            int c                  = (int)( 255.f * rgb[(size_t)y * w * 3 + i] );
            raw[row * y + 1 + i] = (unsigned char)( c < 0 ? 0 : ( c > 255 ? 255 : c ) );
        }
    }
    std::vector<unsigned char> z;
    z.reserve( raw.size() + raw.size() / 65535 * 5 + 16 );
    z.push_back( 0x78 ), z.push_back( 0x01 ); // zlib header: deflate, 32 KiB window, no preset dictionary
    uint32_t a = 1, b = 0;                   // Adler-32 of the raw bytes
    for ( size_t off = 0; off < raw.size(); ) {
        const size_t len = std::min<size_t>( 65535, raw.size() - off );
        z.push_back( off + len == raw.size() ? 1 : 0 ); // BFINAL, BTYPE = 00 (stored)
        z.push_back( (unsigned char)( len & 0xff ) ), z.push_back( (unsigned char)( len >> 8 ) );
        z.push_back( (unsigned char)( ~len & 0xff ) ), z.push_back( (unsigned char)( ( ~len >> 8 ) & 0xff ) );
        z.insert( z.end(), raw.begin() + off, raw.begin() + off + len );
        for ( size_t i = off; i < off + len; i++ ) {
            a = ( a + raw[i] ) % 65521u;
            b = ( b + a ) % 65521u;
        }
        off += len;
    }
    put_be32( z, ( b << 16 ) | a );
    FILE* f = std::fopen( filename, "wb" );
    if ( !f ) return false;
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<unsigned char> ihdr;
    put_be32( ihdr, w ), put_be32( ihdr, h );
    ihdr.push_back( 8 ), ihdr.push_back( 2 ), ihdr.push_back( 0 ), ihdr.push_back( 0 ), ihdr.push_back( 0 ); // 8 bit, RGB, deflate, filter 0, no interlace
    bool ok = std::fwrite( sig, 1, 8, f ) == 8 && write_chunk( f, "IHDR", ihdr ) && write_chunk( f, "IDAT", z ) && write_chunk( f, "IEND", {} );
    return ( std::fclose( f ) == 0 ) && ok;
}

} // namespace mrt
