// parseScene( sceneId ) (example_scenes.cpp:1594-1647) over scene DATA files.
//
// The reference builds its example scenes in C++ (initScene9 .. initScene21) with the asset paths of
// its author's machine compiled in.  Here a scene is a text file, assets/scenes/scene<N>.scene, that
// lists the same calls in the same order -- CreateMaterial, Transform operations, addObject /
// addObjectsFromFile / addLight, setSimpleCamera -- so material, object and light indices come out as
// in the reference, and parseScene( N ) loads that file.  Numbers are parsed as double and narrowed
// to float where the reference assigns a double literal to a float.
//
// Directives (one per line, '#' starts a comment, names are free identifiers):
//   camera from X Y Z at X Y Z up X Y Z fov F lens R focus D
//          focus D: a number, `distance` (glm::distance(from, at)) or `distance times K`
//   env FILE                       new IBL(FILE); addLight; envLights.push_back
//   material NAME [type diffuse|specular|transparent] [albedo R G B] [emission R G B]
//            [roughness X] [metalness X] [ior X]          scene->CreateMaterial(COOK_TORRANCE, type)
//   material NAME model blinn      scene->CreateMaterial(BLINN): a place in the material table (scene 17
//                                  makes one that no object uses); flatten refuses objects that use one
//   texture NAME image FILE | texture NAME checker R G B R G B
//   texture_transform NAME <ops>   the texture's m_transform
//   transform <ops>                the current object transform, built from identity by
//                                  ops = scale SX SY SZ | rotate AX AY AZ DEGREES | translate X Y Z ...
//   plane NAME | sphere NAME       addObject(new Plane / initSphere(material, current transform))
//   mesh FILE NAME                 addObjectsFromFile(FILE, scene, material, current transform)
//   light sphere NAME R G B        addLight(new ShapeLight(.., color, initSphere(material, current transform)))
#include "host_math.hpp"

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace mrt {

namespace {

struct Tokens {
    std::vector<std::string> t;
    size_t i = 0;
    std::string where;
    bool more() const { return i < t.size(); }
    const std::string& peek() const { return t[i]; }
    std::string word() {
        if ( !more() ) throw std::runtime_error( where + ": unexpected end of line" );
        return t[i++];
    }
    double num() {
        std::string w = word();
        char* e       = nullptr;
        double v      = std::strtod( w.c_str(), &e );
        if ( e == w.c_str() || *e ) throw std::runtime_error( where + ": '" + w + "' is not a number" );
        return v;
    }
    vec3 v3() {
        double a = num(), b = num(), c = num();
        return vec3( (float)a, (float)b, (float)c );
    }
    void expect( const char* w ) {
        if ( word() != w ) throw std::runtime_error( where + ": expected '" + w + "'" );
    }
};

void transform_ops( Tokens& tk, Transform& t ) {
    while ( tk.more() ) {
        std::string op = tk.word();
        if ( op == "scale" ) {
            double a = tk.num(), b = tk.num(), c = tk.num();
            t.scale( (float)a, (float)b, (float)c );
        }
        else if ( op == "rotate" ) {
            vec3 axis = tk.v3();
            t.rotate( axis, (float)tk.num() );
        }
        else if ( op == "translate" ) t.translate( tk.v3() );
        else throw std::runtime_error( tk.where + ": unknown transform operation '" + op + "'" );
    }
}

} // namespace

Scene* loadSceneFile( const char* filename, int width, int height, const char* asset_dir ) {
    std::ifstream f( filename );
    if ( !f ) throw std::runtime_error( std::string( "cannot open scene file " ) + filename );
    std::string dir = asset_dir ? asset_dir : "";
    if ( !asset_dir ) { // the scenes live in <assets>/scenes/
        std::string p = filename;
        size_t s      = p.find_last_of( '/' );
        dir           = ( s == std::string::npos ) ? std::string( ".." ) : p.substr( 0, s ) + "/..";
    }
    if ( !dir.empty() && dir.back() != '/' ) dir += '/';

    std::unique_ptr<Scene> scene( initScene() );
    std::map<std::string, size_t> mats;
    auto mat_id = [&]( Tokens& tk ) {
        std::string name = tk.word();
        auto it          = mats.find( name );
        if ( it == mats.end() ) throw std::runtime_error( tk.where + ": unknown material '" + name + "'" );
        return it->second;
    };
    Transform cur;
    std::string line;
    int line_no = 0;
    while ( std::getline( f, line ) ) {
        line_no++;
        size_t hash = line.find( '#' );
        if ( hash != std::string::npos ) line.resize( hash );
        Tokens tk;
        tk.where = std::string( filename ) + ":" + std::to_string( line_no );
        std::istringstream ss( line );
        for ( std::string w; ss >> w; )
            tk.t.push_back( w );
        if ( !tk.more() ) continue;
        std::string cmd = tk.word();
        if ( cmd == "camera" ) {
            tk.expect( "from" );
            vec3 from = tk.v3();
            tk.expect( "at" );
            vec3 at = tk.v3();
            tk.expect( "up" );
            vec3 up = tk.v3();
            tk.expect( "fov" );
            float fov = (float)tk.num();
            tk.expect( "lens" );
            float lens = (float)tk.num();
            tk.expect( "focus" );
            float focus;
            if ( tk.more() && tk.peek() == "distance" ) {
                tk.word();
                float d = length( at - from ); // glm::distance / glm::length( at - from )
                focus   = d;
                if ( tk.more() ) {
                    tk.expect( "times" );
                    focus = (float)( (double)d * tk.num() );
                }
            }
            else focus = (float)tk.num();
            setSimpleCamera( scene.get(), from, at, up, fov, (float)width, (float)height, lens, focus );
        }
        else if ( cmd == "env" ) {
            IBL* sky = new IBL( dir + tk.word() );
            addLight( scene.get(), sky );
            scene->envLights.push_back( sky );
        }
        else if ( cmd == "material" ) {
            std::string name = tk.word();
            if ( mats.count( name ) ) throw std::runtime_error( tk.where + ": material '" + name + "' defined twice" );
            MatType type = DIFFUSE;
            if ( tk.more() && tk.peek() == "model" ) { // a Blinn material only takes its place in the table
                tk.word();
                tk.expect( "blinn" );
                mats[name] = scene->CreateMaterial( MaterialModel::BLINN ).m_UID;
                if ( tk.more() ) throw std::runtime_error( tk.where + ": a Blinn material takes no fields" );
                continue;
            }
            if ( tk.more() && tk.peek() == "type" ) {
                tk.word();
                std::string t = tk.word();
                if ( t == "diffuse" ) type = DIFFUSE;
                else if ( t == "specular" ) type = SPECULAR;
                else if ( t == "transparent" ) type = TRANSPARENT;
                else throw std::runtime_error( tk.where + ": unknown material type '" + t + "'" );
            }
            Material& m = scene->CreateMaterial( MaterialModel::COOK_TORRANCE, type );
            mats[name]  = m.m_UID;
            while ( tk.more() ) {
                std::string k = tk.word();
                if ( k == "albedo" ) m.m_albedo = tk.v3();
                else if ( k == "emission" ) m.m_emission = tk.v3();
                else if ( k == "roughness" ) m.m_roughness = (float)tk.num();
                else if ( k == "metalness" ) m.m_metalness = (float)tk.num();
                else if ( k == "ior" ) m.m_IOR = (float)tk.num();
                else throw std::runtime_error( tk.where + ": unknown material field '" + k + "'" );
            }
        }
        else if ( cmd == "texture" ) {
            size_t id        = mat_id( tk );
            std::string kind = tk.word();
            texture* t       = nullptr;
            if ( kind == "image" ) {
                std::string file = dir + tk.word();
                t                = image_texture( file.c_str() );
                if ( !t ) throw std::runtime_error( tk.where + ": cannot load texture " + file );
            }
            else if ( kind == "checker" ) {
                vec3 a = tk.v3();
                t      = checker_texture( a, tk.v3() );
            }
            else throw std::runtime_error( tk.where + ": unknown texture kind '" + kind + "'" );
            scene->owned_textures.push_back( t );
            scene->GetMaterial( (int)id ).m_texture = t;
        }
        else if ( cmd == "texture_transform" ) {
            size_t id  = mat_id( tk );
            texture* t = scene->GetMaterial( (int)id ).m_texture;
            if ( !t ) throw std::runtime_error( tk.where + ": the material has no texture" );
            Transform x;
            transform_ops( tk, x );
            t->m_transform = x;
        }
        else if ( cmd == "transform" ) {
            cur = Transform();
            transform_ops( tk, cur );
        }
        else if ( cmd == "plane" ) addObject( scene.get(), new Plane( mat_id( tk ), cur ) );
        else if ( cmd == "sphere" ) addObject( scene.get(), initSphere( (int)mat_id( tk ), cur ) );
        else if ( cmd == "mesh" ) {
            std::string file = dir + tk.word();
            size_t id        = mat_id( tk );
            addObjectsFromFileEx( file.c_str(), scene.get(), id, cur, 1.f, 0 );
        }
        else if ( cmd == "light" ) {
            tk.expect( "sphere" );
            size_t id = mat_id( tk );
            vec3 c    = tk.v3();
            addLight( scene.get(), new ShapeLight( vec3( 0.f ), c, initSphere( (int)id, cur ) ) );
        }
        else throw std::runtime_error( tk.where + ": unknown directive '" + cmd + "'" );
        if ( tk.more() ) throw std::runtime_error( tk.where + ": trailing '" + tk.peek() + "'" );
    }
    if ( !scene->cam ) throw std::runtime_error( std::string( filename ) + ": no camera" );
    return scene.release();
}

Scene* parseScene( int sceneId, int width, int height, const char* asset_dir ) {
    std::string dir;
    if ( asset_dir ) dir = asset_dir;
    else if ( const char* e = std::getenv( "MRT_ASSETS" ) ) dir = e;
    else dir = "../assets"; // where the reference looks for its MTL files (scene.cpp:187)
    if ( !dir.empty() && dir.back() != '/' ) dir += '/';
    switch ( sceneId ) {
    default: // example_scenes.cpp:1637-1639: an unknown id gives scene 19
        sceneId = 19;
        // fall through
    case 9:
    case 16:
    case 17:
    case 18:
    case 19:
    case 20:
    case 21:
        return loadSceneFile( ( dir + "scenes/scene" + std::to_string( sceneId ) + ".scene" ).c_str(), width, height, dir.c_str() );
    case 10:
    case 11:
    case 12:
    case 13:
    case 14:
    case 15:
        // Blinn materials under point / directional / ambient lights, written for the Whitted
        // integrator that the reference has switched off; its Pathtracer evaluates a Blinn material
        // to black (Material.cpp:1011, 1098), so there is nothing for this backend to render.
        throw std::invalid_argument( "parseScene: scene " + std::to_string( sceneId ) + " uses the Blinn material model, which the path tracer renders black" );
    }
}

} // namespace mrt
