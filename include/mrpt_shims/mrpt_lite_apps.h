/*
 * mrpt_lite_apps.h -- stand-ins for the MRPT application-level classes that the reference's tutorials and srba-slam front-end touch around the RbaEngine<> calls
 * (3D scene containers, the display window, file streams, command-line parsing helpers). They exist so that such code COMPILES against this repo where MRPT is not
 * installed; nothing in the numeric path uses them. Scene objects record the primitives they are given (so a caller can inspect or export them); windows do nothing.
 * A build against the real MRPT does not put include/mrpt_shims on the include path.
 */
#pragma once
#include <mrpt_lite.h>
#include <chrono>
#include <ctime>
#include <list>
#include <memory>
#include <set>

namespace mrpt {
namespace utils {
struct TColorf { float R, G, B, A; TColorf(float r = 1, float g = 1, float b = 1, float a = 1) : R(r), G(g), B(b), A(a) {} };
struct TColor { unsigned char R, G, B, A; TColor(unsigned char r = 0, unsigned char g = 0, unsigned char b = 0, unsigned char a = 255) : R(r), G(g), B(b), A(a) {} };
typedef std::map<std::string, double> TParametersDouble;
} // namespace utils

namespace opengl {
/** smart pointer with the MRPT 1.x spelling (present() / clear_unique()) */
template <class T> struct ptr : public std::shared_ptr<T> {
	ptr() {} ptr(const std::shared_ptr<T> &p) : std::shared_ptr<T>(p) {} template <class U> ptr(const ptr<U> &p) : std::shared_ptr<T>(std::static_pointer_cast<T>(std::shared_ptr<U>(p))) {}
	bool present() const { return (bool)*this; } void clear_unique() { this->reset(); } T *pointer() { return this->get(); }
};
struct CRenderizable {
	virtual ~CRenderizable() {}
	std::string m_name; mrpt::poses::CPose3D m_pose; utils::TColorf m_color; bool m_visible = true;
	CRenderizable &setName(const std::string &n) { m_name = n; return *this; } CRenderizable &setPose(const mrpt::poses::CPose3D &p) { m_pose = p; return *this; }
	CRenderizable &setColor(double r, double g, double b, double a = 1) { m_color = utils::TColorf((float)r, (float)g, (float)b, (float)a); return *this; }
		CRenderizable &setColor(const utils::TColorf &c) { m_color = c; return *this; }
	CRenderizable &setVisibility(bool v = true) { m_visible = v; return *this; } CRenderizable &enableShowName(bool = true) { return *this; } CRenderizable &setLocation(double x, double y,
		double z) { m_pose.m_t[0] = x; m_pose.m_t[1] = y; m_pose.m_t[2] = z; return *this; }
};
typedef ptr<CRenderizable> CRenderizablePtr;
#define SRBA_LITE_GL_CLASS(NAME) struct NAME; typedef ptr<NAME> NAME##Ptr
SRBA_LITE_GL_CLASS(CSetOfObjects); SRBA_LITE_GL_CLASS(CSetOfLines); SRBA_LITE_GL_CLASS(CPointCloud); SRBA_LITE_GL_CLASS(CGridPlaneXY); SRBA_LITE_GL_CLASS(CText); SRBA_LITE_GL_CLASS(COpenGLScene);
	SRBA_LITE_GL_CLASS(COpenGLViewport); SRBA_LITE_GL_CLASS(CCamera);
struct CSetOfLines : CRenderizable {
	struct seg { double x0, y0, z0, x1, y1, z1; }; std::vector<seg> segments; float width = 1;
	static CSetOfLinesPtr Create() { return CSetOfLinesPtr(std::make_shared<CSetOfLines>()); }
	void appendLine(double x0, double y0, double z0, double x1, double y1, double z1) { const seg s = {x0, y0, z0, x1, y1, z1}; segments.push_back(s); }
	void appendLineStrip(double x, double y, double z) { if (has_last) appendLine(lx, ly, lz, x, y, z); lx = x; ly = y; lz = z; has_last = true; }
	void setLineWidth(float w) { width = w; } void clear() { segments.clear(); has_last = false; } size_t size() const { return segments.size(); }
private:
	double lx = 0, ly = 0, lz = 0; bool has_last = false;
};
struct CPointCloud : CRenderizable {
	std::vector<double> xs, ys, zs; float psize = 1;
	static CPointCloudPtr Create() { return CPointCloudPtr(std::make_shared<CPointCloud>()); }
	void insertPoint(double x, double y, double z) { xs.push_back(x); ys.push_back(y); zs.push_back(z); } void setPointSize(float s) { psize = s; } void clear() { xs.clear(); ys.clear(); zs.clear();
		} size_t size() const { return xs.size(); }
	void resize(size_t n) { xs.resize(n); ys.resize(n); zs.resize(n); } void setPoint_fast(size_t i, double x, double y, double z) { xs[i] = x; ys[i] = y; zs[i] = z; } void setPoint(size_t i,
		double x, double y, double z) { setPoint_fast(i, x, y, z); }
	void getBoundingBox(mrpt::math::TPoint3D &lo, mrpt::math::TPoint3D &hi) const { lo = hi = mrpt::math::TPoint3D(); for (size_t i = 0; i < xs.size(); i++) { if (!i || xs[i] < lo.x) lo.x = xs[i];
		if (!i || ys[i] < lo.y) lo.y = ys[i]; if (!i || zs[i] < lo.z) lo.z = zs[i]; if (!i || xs[i] > hi.x) hi.x = xs[i]; if (!i || ys[i] > hi.y) hi.y = ys[i]; if (!i || zs[i] > hi.z) hi.z = zs[i]; }
		}
};
struct CGridPlaneXY : CRenderizable { static CGridPlaneXYPtr Create(float = -10, float = 10, float = -10, float = 10, float = 0, float = 1) { return CGridPlaneXYPtr(std::make_shared<CGridPlaneXY>());
	} };
struct CText : CRenderizable { std::string text; static CTextPtr Create(const std::string &s = "") { CTextPtr t(std::make_shared<CText>()); t->text = s; return t; }
	void setString(const std::string &s) { text = s; } };
/** container of scene objects; also the sink of RbaEngine<>::build_opengl_representation() (insert_corner / insert_line / insert_point / insert_text) */
struct CSetOfObjects : CRenderizable {
	std::vector<CRenderizablePtr> objects;
	static CSetOfObjectsPtr Create() { return CSetOfObjectsPtr(std::make_shared<CSetOfObjects>()); }
	template <class P> void insert(const P &o) { objects.push_back(CRenderizablePtr(o)); } void clear() { objects.clear(); m_lines.reset(); m_pts[0].reset(); m_pts[1].reset(); } size_t size() const {
		return objects.size(); } bool empty() const { return objects.empty(); }
	/** extent of the recorded primitives (poses of the children, line end points, points) */
	void getBoundingBox(mrpt::math::TPoint3D &lo, mrpt::math::TPoint3D &hi) const;
	void insert_corner(const mrpt::poses::CPose3D &p, double scale);
	void insert_line(const mrpt::poses::CPose3D &a, const mrpt::poses::CPose3D &b) { if (!m_lines) { m_lines = CSetOfLines::Create(); objects.push_back(CRenderizablePtr(m_lines)); }
		m_lines->appendLine(a.x(), a.y(), a.z(), b.x(), b.y(), b.z()); }
	void insert_point(double x, double y, double z, bool unknown) { CPointCloudPtr &pc = m_pts[unknown ? 1 : 0]; if (!pc) { pc = CPointCloud::Create();
		pc->setName(unknown ? "unknown landmarks" : "fixed landmarks"); objects.push_back(CRenderizablePtr(pc)); } pc->insertPoint(x, y, z); }
	void insert_text(const mrpt::poses::CPose3D &p, const std::string &s) { CTextPtr t = CText::Create(s); t->setPose(p); objects.push_back(CRenderizablePtr(t)); }
private:
	CSetOfLinesPtr m_lines; CPointCloudPtr m_pts[2];
};
namespace stock_objects {
inline CSetOfObjectsPtr CornerXYZSimple(float scale = 1, float = 1) { CSetOfObjectsPtr o = CSetOfObjects::Create(); o->setName(mrpt::format("corner %.3f", scale)); return o; }
inline CSetOfObjectsPtr CornerXYZ(float scale = 1) { return CornerXYZSimple(scale); }
inline CSetOfObjectsPtr CornerXYSimple(float scale = 1, float = 1) { return CornerXYZSimple(scale); }
} // namespace stock_objects
inline void CSetOfObjects::getBoundingBox(mrpt::math::TPoint3D &lo, mrpt::math::TPoint3D &hi) const {
	bool any = false; lo = hi = mrpt::math::TPoint3D();
	auto take = [&](double x, double y, double z) { if (!any) { lo = hi = mrpt::math::TPoint3D(x, y, z); any = true; return; } lo.x = std::min(lo.x, x); lo.y = std::min(lo.y, y);
		lo.z = std::min(lo.z, z); hi.x = std::max(hi.x, x); hi.y = std::max(hi.y, y); hi.z = std::max(hi.z, z); };
	for (size_t i = 0; i < objects.size(); i++) {
		const CRenderizable *o = objects[i].get(); take(o->m_pose.x(), o->m_pose.y(), o->m_pose.z());
		if (const CSetOfLines *l = dynamic_cast<const CSetOfLines *>(o)) for (size_t k = 0; k < l->segments.size(); k++) { take(l->segments[k].x0, l->segments[k].y0, l->segments[k].z0);
			take(l->segments[k].x1, l->segments[k].y1, l->segments[k].z1); }
		if (const CPointCloud *pc = dynamic_cast<const CPointCloud *>(o)) for (size_t k = 0; k < pc->xs.size(); k++) take(pc->xs[k], pc->ys[k], pc->zs[k]);
	}
}
inline void CSetOfObjects::insert_corner(const mrpt::poses::CPose3D &p, double scale) { CSetOfObjectsPtr c = stock_objects::CornerXYZSimple((float)scale); c->setPose(p);
	objects.push_back(CRenderizablePtr(c)); }
enum TOpenGLFontStyle { FILL = 0, OUTLINE = 1, NICE = 2 };
struct CCamera : CRenderizable { void setOrthogonal(bool = true) {} void setZoomDistance(float) {} void setAzimuthDegrees(float) {} void setElevationDegrees(float) {} void setPointingAt(double,
	double, double) {} void setProjectiveModel(bool) {} };
struct COpenGLViewport {
	std::vector<CRenderizablePtr> objects; CCamera cam;
	void setViewportPosition(double, double, double, double) {} void setCloneView(const std::string &) {} void setTransparent(bool) {} void setBorderSize(unsigned) {}
		void setCustomBackgroundColor(const utils::TColorf &) {}
	template <class P> void insert(const P &o) { objects.push_back(CRenderizablePtr(o)); } void clear() { objects.clear(); } CCamera &getCamera() { return cam; }
};
struct COpenGLScene {
	std::map<std::string, COpenGLViewportPtr> views;
	static COpenGLScenePtr Create() { return COpenGLScenePtr(std::make_shared<COpenGLScene>()); }
	COpenGLViewportPtr getViewport(const std::string &n = "main") { COpenGLViewportPtr &v = views[n]; if (!v) v = COpenGLViewportPtr(std::make_shared<COpenGLViewport>()); return v; }
	COpenGLViewportPtr createViewport(const std::string &n) { return getViewport(n); }
	template <class P> void insert(const P &o, const std::string &view = "main") { getViewport(view)->insert(o); } void clear() { views.clear(); }
	bool saveToFile(const std::string &) const { return false; }
};
namespace graph_tools {
template <class GRAPH> CSetOfObjectsPtr graph_visualize(const GRAPH &g, const utils::TParametersDouble & = utils::TParametersDouble()) {
	CSetOfObjectsPtr o = CSetOfObjects::Create();
	for (typename GRAPH::global_poses_t::const_iterator it = g.nodes.begin(); it != g.nodes.end(); ++it) o->insert_corner(mrpt::poses::CPose3D(it->second), 0.25);
	return o;
}
} // namespace graph_tools
} // namespace opengl

namespace utils {
struct TImageSize { unsigned x = 0, y = 0; };
struct CImage { unsigned w = 0, h = 0; unsigned getWidth() const { return w; } unsigned getHeight() const { return h; } TImageSize getSize() const { TImageSize s; s.x = w; s.y = h; return s; }
	bool saveToFile(const std::string &) const { return false; } };
typedef opengl::ptr<CImage> CImagePtr;
/** stopwatch (mrpt::utils::CTicTac) */
class CTicTac { std::chrono::steady_clock::time_point t0; public: CTicTac() { Tic(); } void Tic() { t0 = std::chrono::steady_clock::now(); } double Tac() const {
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } };
/** text output file with printf (mrpt::utils::CFileOutputStream) */
class CFileOutputStream {
	FILE *f;
public:
	explicit CFileOutputStream(const std::string &file, bool append = false) : f(fopen(file.c_str(), append ? "ab" : "wb")) { if (!f) throw std::runtime_error("cannot open for writing: " + file); }
	~CFileOutputStream() { if (f) fclose(f); } CFileOutputStream(const CFileOutputStream &) = delete;
	int printf(const char *fmt, ...) __attribute__((format(printf, 2, 3))) { va_list ap; va_start(ap, fmt); const int r = vfprintf(f, fmt, ap); va_end(ap); return r; }
	bool fileOpenCorrectly() const { return f != NULL; } void close() { if (f) { fclose(f); f = NULL; } }
};
/** binary cache files of the dataset parsers: plain (uncompressed) streams of the matrices' raw contents; anything else streamed into them is dropped */
class CFileGZOutputStream {
	std::ofstream f;
public:
	explicit CFileGZOutputStream(const std::string &file) : f(file.c_str(), std::ios::binary) { if (!f) throw std::runtime_error("cannot open for writing: " + file); }
	CFileGZOutputStream &operator<<(const mrpt::math::CMatrixDouble &M) { const uint64_t hdr[2] = {M.nr, M.nc}; f.write((const char *)hdr, sizeof(hdr));
		if (!M.m.empty()) f.write((const char *)&M.m[0], sizeof(double) * M.m.size()); return *this; }
	CFileGZOutputStream &operator<<(const std::vector<mrpt::poses::CPose3DQuat> &v) { const uint64_t n = v.size(); f.write((const char *)&n, sizeof(n)); for (size_t i = 0; i < v.size(); i++) {
		const double r[7] = {v[i].x(), v[i].y(), v[i].z(), v[i].quat().r(), v[i].quat().x(), v[i].quat().y(), v[i].quat().z()}; f.write((const char *)r, sizeof(r)); } return *this; }
	template <class T> CFileGZOutputStream &operator<<(const T &) { return *this; }
};
class CFileGZInputStream {
	std::ifstream f;
public:
	explicit CFileGZInputStream(const std::string &file) : f(file.c_str(), std::ios::binary) { if (!f) throw std::runtime_error("cannot open: " + file); }
	CFileGZInputStream &operator>>(mrpt::math::CMatrixDouble &M) { uint64_t hdr[2] = {0, 0}; f.read((char *)hdr, sizeof(hdr)); M.setSize(hdr[0], hdr[1]); if (!M.m.empty()) f.read((char *)&M.m[0],
		sizeof(double) * M.m.size()); if (!f) throw std::runtime_error("truncated matrix stream"); return *this; }
	CFileGZInputStream &operator>>(std::vector<mrpt::poses::CPose3DQuat> &v) { uint64_t n = 0; f.read((char *)&n, sizeof(n)); v.clear(); for (uint64_t i = 0; i < n; i++) { double r[7];
		f.read((char *)r, sizeof(r)); v.push_back(mrpt::poses::CPose3DQuat(r[0], r[1], r[2], mrpt::math::CQuaternionDouble(r[3], r[4], r[5], r[6]))); }
		if (!f) throw std::runtime_error("truncated pose stream"); return *this; }
};
/** line-by-line reader that skips blank and comment lines (mrpt::utils::CTextFileLinesParser) */
class CTextFileLinesParser {
	std::ifstream f; size_t m_line = 0;
public:
	explicit CTextFileLinesParser(const std::string &file) : f(file.c_str()) { if (!f) throw std::runtime_error("cannot open: " + file); }
	bool getNextLine(std::string &out) { while (std::getline(f, out)) { m_line++; const size_t a = out.find_first_not_of(" \t\r"); if (a == std::string::npos) continue;
		if (out[a] == '#' || out[a] == '%' || out.compare(a, 2, "//") == 0) continue; return true; } return false; }
	bool getNextLine(std::istringstream &out) { std::string s; if (!getNextLine(s)) return false; out.clear(); out.str(s); return true; }
	size_t getCurrentLineNumber() const { return m_line; }
};
} // namespace utils

namespace graphs {
/** pose graph container (mrpt::graphs::CNetworkOfPoses): what RbaEngine<>::get_global_graphslam_problem() fills */
template <class POSE> struct CNetworkOfPoses {
	typedef POSE constraint_t; typedef std::map<uint64_t, POSE> global_poses_t; typedef std::multimap<std::pair<uint64_t, uint64_t>, POSE> edges_map_t;
		typedef typename edges_map_t::const_iterator const_iterator;
	global_poses_t nodes; edges_map_t edges; uint64_t root = 0;
	void clear() { nodes.clear(); edges.clear(); root = 0; }
	void insertEdgeAtEnd(uint64_t from, uint64_t to, const POSE &p) { edges.insert(edges.end(), std::make_pair(std::make_pair(from, to), p)); }
	void insertEdge(uint64_t from, uint64_t to, const POSE &p) { edges.insert(std::make_pair(std::make_pair(from, to), p)); }
	size_t nodeCount() const { return nodes.size(); } size_t edgeCount() const { return edges.size(); } const_iterator begin() const { return edges.begin(); } const_iterator end() const {
		return edges.end(); }
};
typedef CNetworkOfPoses<mrpt::poses::CPose2D> CNetworkOfPoses2D; typedef CNetworkOfPoses<mrpt::poses::CPose3D> CNetworkOfPoses3D;
} // namespace graphs
namespace graphslam {
struct TResultInfoSpaLevMarq { size_t num_iters = 0; double final_total_sq_error = 0; };
/** global pose-graph optimisation is MRPT's, not part of SRBA: absent from this stand-in */
template <class GRAPH> void optimize_graph_spa_levmarq(GRAPH &, TResultInfoSpaLevMarq &, const std::set<uint64_t> * = NULL, const utils::TParametersDouble & = utils::TParametersDouble()) {
	throw std::runtime_error("mrpt::graphslam::optimize_graph_spa_levmarq needs the real MRPT"); }
} // namespace graphslam

namespace vision {
/** no video encoder in this build */
class CVideoFileWriter { public: bool open(const std::string &, double, const utils::TImageSize &, const std::string & = "", bool = true) { return false; } bool isOpen() const { return false; }
	void close() {} template <class IMG> const CVideoFileWriter &operator<<(const IMG &) const { return *this; } };
} // namespace vision

namespace gui {
class CDisplayWindow3D; typedef opengl::ptr<CDisplayWindow3D> CDisplayWindow3DPtr;
/** no display in this build: the window accepts the calls and shows nothing */
class CDisplayWindow3D {
public:
	static CDisplayWindow3DPtr Create(const std::string &caption = "", unsigned w = 640, unsigned h = 480) { return CDisplayWindow3DPtr(std::make_shared<CDisplayWindow3D>(caption, w, h)); }
	CDisplayWindow3D(const std::string & = "", unsigned = 640, unsigned = 480) : m_scene(opengl::COpenGLScene::Create()) {}
	opengl::COpenGLScenePtr &get3DSceneAndLock() { return m_scene; } void unlockAccess3DScene() {} void repaint() {} void forceRepaint() {}
	void setCameraZoom(float) {} void setCameraAzimuthDeg(float) {} void setCameraElevationDeg(float) {} void setCameraPointingToPoint(float, float, float) {} void setPos(int, int) {}
		void resize(unsigned, unsigned) {}
	bool isOpen() const { return false; } bool keyHit() const { return false; } int waitForKey(bool = true) { return 0; } int getPushedKey() { return 0; } void clearKeyHitFlag() {}
	void captureImagesStart() {} void captureImagesStop() {} utils::CImagePtr getLastWindowImagePtr() { return utils::CImagePtr(); } void grabImagesStart(const std::string & = "") {}
		void grabImagesStop() {}
	template <class... A> void addTextMessage(double, double, const std::string &, A...) {}
private:
	opengl::COpenGLScenePtr m_scene;
};
} // namespace gui

namespace system {
inline bool fileExists(const std::string &f) { std::ifstream s(f.c_str()); return s.good(); }
inline bool directoryExists(const std::string &) { return true; }
inline std::string extractFileName(const std::string &p) { const size_t a = p.find_last_of("/\\"), b = p.find_last_of('.'); const size_t s = a == std::string::npos ? 0 : a + 1; return p.substr(s,
	(b == std::string::npos || b < s) ? std::string::npos : b - s); }
inline std::string extractFileExtension(const std::string &p) { const size_t b = p.find_last_of('.'); return b == std::string::npos ? std::string() : p.substr(b + 1); }
inline std::string extractFileDirectory(const std::string &p) { const size_t a = p.find_last_of("/\\"); return a == std::string::npos ? std::string() : p.substr(0, a + 1); }
inline std::string fileNameChangeExtension(const std::string &p, const std::string &ext) { const size_t b = p.find_last_of('.'); return (b == std::string::npos ? p : p.substr(0, b)) + "." + ext; }
inline std::string fileNameStripInvalidChars(const std::string &p) { return p; }
inline bool createDirectory(const std::string &) { return true; } inline bool deleteFilesInDirectory(const std::string &, bool = false) { return true; }
inline void sleep(int ms); inline void pause() {}
typedef uint64_t TTimeStamp;
inline TTimeStamp now() { return (TTimeStamp)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); } inline TTimeStamp getCurrentTime() {
	return now(); } inline TTimeStamp getCurrentLocalTime() { return now(); }
inline std::string dateTimeToString(TTimeStamp t) { const time_t s = (time_t)(t / 1000000); char buf[64]; struct tm tmv; gmtime_r(&s, &tmv); strftime(buf, sizeof(buf), "%Y/%m/%d,%H:%M:%S", &tmv);
	return std::string(buf); }
inline std::string formatTimeInterval(double seconds) { return mrpt::format("%02u:%02u:%02u", (unsigned)(seconds / 3600), (unsigned)(seconds / 60) % 60, (unsigned)seconds % 60); }
time_t getFileModificationTime(const std::string &file);
inline std::string MRPT_getVersion() { return "mrpt_lite"; }
namespace os { inline bool kbhit() { return false; } inline int getch() { return 0; } }
inline void setConsoleColor(int) {}
enum { CONCOL_NORMAL = 0, CONCOL_BLUE, CONCOL_GREEN, CONCOL_RED };
} // namespace system
} // namespace mrpt
#include <sys/stat.h>
#include <thread>
inline time_t mrpt::system::getFileModificationTime(const std::string &file) { struct stat st; return stat(file.c_str(), &st) == 0 ? st.st_mtime : (time_t)0; }
inline void mrpt::system::sleep(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }
#define ASSERT_FILE_EXISTS_(f) ASSERTMSG_(mrpt::system::fileExists(f), std::string("File not found: ") + (f))
