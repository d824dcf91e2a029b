// On-disk formats either side of the accelerated path (host only):
//   * CameraCalibration::Data::save / load text file ("calib.data")
//       /root/reference/modules/calib/src/calibration.cpp:22-79
//   * PinholeCamera::read / write YAML (cv::Mat as {rows, cols, data})
//       /root/reference/modules/util/include/cvx/util/camera/camera.hpp:19-72,138-180
//   * CameraRobotConfiguration YAML + the static-transform launch file it writes
//       /root/reference/modules/util/include/cvx/util/misc/configuration.hpp:31-91
// The reference reads YAML with yaml-cpp (un-vendored); the files involved use a small subset -- nested block
// maps, block or flow sequences of scalars, plain or quoted scalars, comments -- which the parser below
// covers.  Numbers are written the way the reference's streams write them: iostream defaults (6 significant
// digits) for the text formats, 17 significant digits for YAML doubles as yaml-cpp does.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"

// ------------------------------------------------------------------------------------------
// calib.data
// ------------------------------------------------------------------------------------------
struct cvx_calib_data {
    int width = 0, height = 0;
    std::vector<float> coords;                  // n_coords * 3
    std::vector<std::string> paths;             // n_views
    std::vector<int> view_off, point_id;
    std::vector<float> px, py;
};

extern "C" int cvx_calib_data_load(const char *path, cvx_calib_data **out) {
    if (!path || !out) return CVX_ERR_INVALID_ARG;
    *out = nullptr;
    std::ifstream strm(path);
    if (!strm) { cvx::set_last_error(std::string("cannot open ") + path); return CVX_ERR_INVALID_ARG; }
    cvx_calib_data *d = new cvx_calib_data;
    size_t n_coords = 0, n_views = 0;
    strm >> d->width >> d->height >> n_coords;
    if (!strm || n_coords > (1u << 24)) { delete d; cvx::set_last_error("calib data: bad header"); return CVX_ERR_INVALID_ARG; }
    d->coords.resize(3 * n_coords);
    for (size_t i = 0; i < 3 * n_coords; ++i) strm >> d->coords[i];
    strm >> n_views;
    if (!strm || n_views > (1u << 24)) { delete d; cvx::set_last_error("calib data: bad point list"); return CVX_ERR_INVALID_ARG; }
    d->paths.resize(n_views);
    d->view_off.assign(n_views + 1, 0);
    for (size_t i = 0; i < n_views; ++i) {
        size_t n_markers = 0;
        strm >> d->paths[i] >> n_markers;
        if (!strm || n_markers > (1u << 24)) { delete d; cvx::set_last_error("calib data: bad view record"); return CVX_ERR_INVALID_ARG; }
        for (size_t j = 0; j < n_markers; ++j) {
            float x, y;
            int idx;
            strm >> x >> y >> idx;
            d->px.push_back(x); d->py.push_back(y); d->point_id.push_back(idx);
        }
        if (!strm) { delete d; cvx::set_last_error("calib data: truncated marker list"); return CVX_ERR_INVALID_ARG; }
        d->view_off[i + 1] = (int)d->point_id.size();
    }
    *out = d;
    return CVX_OK;
}

extern "C" void cvx_calib_data_free(cvx_calib_data *d) { delete d; }

extern "C" int cvx_calib_data_info(const cvx_calib_data *d, int *width, int *height, int *n_coords, int *n_views,
                                   int64_t *n_obs) {
    if (!d) return CVX_ERR_INVALID_ARG;
    if (width) *width = d->width;
    if (height) *height = d->height;
    if (n_coords) *n_coords = (int)(d->coords.size() / 3);
    if (n_views) *n_views = (int)d->paths.size();
    if (n_obs) *n_obs = (int64_t)d->point_id.size();
    return CVX_OK;
}

extern "C" int cvx_calib_data_copy(const cvx_calib_data *d, double *coords, int *view_off, int *point_id, double *obs_x,
                                   double *obs_y) {
    if (!d) return CVX_ERR_INVALID_ARG;
    if (coords) for (size_t i = 0; i < d->coords.size(); ++i) coords[i] = (double)d->coords[i];   // Point3f -> double
    if (view_off) std::memcpy(view_off, d->view_off.data(), sizeof(int) * d->view_off.size());
    if (point_id) std::memcpy(point_id, d->point_id.data(), sizeof(int) * d->point_id.size());
    if (obs_x) for (size_t i = 0; i < d->px.size(); ++i) obs_x[i] = (double)d->px[i];              // Point2f -> double
    if (obs_y) for (size_t i = 0; i < d->py.size(); ++i) obs_y[i] = (double)d->py[i];
    return CVX_OK;
}

extern "C" const char *cvx_calib_data_path(const cvx_calib_data *d, int view) {
    if (!d || view < 0 || view >= (int)d->paths.size()) return "";
    return d->paths[view].c_str();
}

extern "C" int cvx_calib_data_save(const char *path, int width, int height, int n_coords, const double *coords,
                                   int n_views, const char *const *image_paths, const int *view_off,
                                   const int *point_id, const double *obs_x, const double *obs_y) {
    if (!path || n_coords < 0 || n_views < 0 || (n_coords && !coords) || (n_views && (!view_off || !image_paths)))
        return CVX_ERR_INVALID_ARG;
    std::ofstream strm(path);
    if (!strm) { cvx::set_last_error(std::string("cannot write ") + path); return CVX_ERR_INVALID_ARG; }
    strm << width << ' ' << height << std::endl;
    strm << (size_t)n_coords << std::endl;
    for (int i = 0; i < n_coords; ++i)      // cv::Point3f members: floats at the stream's default precision
        strm << (float)coords[3 * i] << ' ' << (float)coords[3 * i + 1] << ' ' << (float)coords[3 * i + 2] << std::endl;
    strm << (size_t)n_views << std::endl;
    for (int v = 0; v < n_views; ++v) {
        strm << image_paths[v] << std::endl;
        strm << (size_t)(view_off[v + 1] - view_off[v]) << std::endl;
        for (int o = view_off[v]; o < view_off[v + 1]; ++o)
            strm << (float)obs_x[o] << ' ' << (float)obs_y[o] << ' ' << point_id[o] << std::endl;
    }
    return strm ? CVX_OK : CVX_ERR_INVALID_ARG;
}

// ------------------------------------------------------------------------------------------
// YAML subset
// ------------------------------------------------------------------------------------------
namespace {

struct YNode {
    std::string scalar;
    std::vector<std::string> seq;
    std::map<std::string, YNode> map;
    bool is_seq = false;
};

std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

std::string unquote(std::string s) {
    s = trim(s);
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

std::string strip_comment(const std::string &line) {
    bool dq = false, sq = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (c == '"' && !sq) dq = !dq;
        else if (c == '\'' && !dq) sq = !sq;
        else if (c == '#' && !dq && !sq && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) return line.substr(0, i);
    }
    return line;
}

void parse_flow_seq(const std::string &v, YNode &n) {
    n.is_seq = true;
    std::string inner = v.substr(1, v.rfind(']') - 1), item;
    std::stringstream ss(inner);
    while (std::getline(ss, item, ',')) if (!trim(item).empty()) n.seq.push_back(unquote(item));
}

struct Line { int indent; std::string text; };

// parses lines[i..) whose indent is >= `indent` into node (a block map or a block sequence)
size_t parse_block(const std::vector<Line> &lines, size_t i, int indent, YNode &node) {
    while (i < lines.size() && lines[i].indent >= indent) {
        const Line &ln = lines[i];
        if (ln.indent > indent) { ++i; continue; }      // stray deeper line: skipped
        if (ln.text[0] == '-' && (ln.text.size() == 1 || ln.text[1] == ' ')) {
            node.is_seq = true;
            node.seq.push_back(unquote(ln.text.substr(1)));
            ++i;
            continue;
        }
        const size_t colon = ln.text.find(':');
        if (colon == std::string::npos) { ++i; continue; }
        const std::string key = unquote(ln.text.substr(0, colon)), val = trim(ln.text.substr(colon + 1));
        YNode child;
        ++i;
        if (val.empty()) {
            if (i < lines.size() && (lines[i].indent > indent ||
                                     (lines[i].indent == indent && lines[i].text[0] == '-')))
                i = parse_block(lines, i, lines[i].indent, child);
        } else if (val[0] == '[') {
            parse_flow_seq(val, child);
        } else {
            child.scalar = unquote(val);
        }
        node.map[key] = child;
    }
    return i;
}

bool load_yaml(const char *path, YNode &root) {
    std::ifstream f(path);
    if (!f) { cvx::set_last_error(std::string("cannot open ") + path); return false; }
    std::vector<Line> lines;
    std::string raw;
    while (std::getline(f, raw)) {
        std::string s = strip_comment(raw);
        if (trim(s).empty() || trim(s) == "---") continue;
        const size_t ind = s.find_first_not_of(' ');
        lines.push_back({(int)ind, trim(s)});
    }
    parse_block(lines, 0, lines.empty() ? 0 : lines[0].indent, root);
    return true;
}

bool get_double(const YNode &n, const char *key, double &out) {
    auto it = n.map.find(key);
    if (it == n.map.end() || it->second.scalar.empty()) { cvx::set_last_error(std::string("missing key ") + key); return false; }
    char *end = nullptr;
    out = std::strtod(it->second.scalar.c_str(), &end);
    if (end == it->second.scalar.c_str()) { cvx::set_last_error(std::string("not a number: ") + key); return false; }
    return true;
}

// camera.hpp:45-66: data in row-major order, rows x cols
bool get_mat(const YNode &n, const char *key, int &rows, int &cols, std::vector<double> &data) {
    auto it = n.map.find(key);
    if (it == n.map.end()) { cvx::set_last_error(std::string("missing key ") + key); return false; }
    double r, c;
    if (!get_double(it->second, "rows", r) || !get_double(it->second, "cols", c)) return false;
    rows = (int)r; cols = (int)c;
    auto d = it->second.map.find("data");
    if (d == it->second.map.end() || !d->second.is_seq) { cvx::set_last_error(std::string("missing data of ") + key); return false; }
    data.clear();
    for (const std::string &s : d->second.seq) data.push_back(std::strtod(s.c_str(), nullptr));
    return true;
}

std::string yaml_double(double v) {      // yaml-cpp: stream precision max_digits10
    if (std::isnan(v)) return ".nan";
    if (std::isinf(v)) return v > 0 ? ".inf" : "-.inf";
    std::stringstream ss;
    ss.precision(17);
    ss << v;
    return ss.str();
}

void set_str(char *dst, size_t cap, const std::string &s) {
    if (!dst || !cap) return;
    std::snprintf(dst, cap, "%s", s.c_str());
}

}  // namespace

// PinholeCamera::read (camera.hpp:138-162): image_width, image_height, camera_matrix, distortion_coefficients
extern "C" int cvx_camera_yaml_read(const char *path, int *width, int *height, double K[4], double dist[5]) {
    if (!path) return CVX_ERR_INVALID_ARG;
    YNode root;
    if (!load_yaml(path, root)) return CVX_ERR_INVALID_ARG;
    double w, h;
    if (!get_double(root, "image_width", w) || !get_double(root, "image_height", h)) return CVX_ERR_INVALID_ARG;
    int r, c;
    std::vector<double> m, d;
    if (!get_mat(root, "camera_matrix", r, c, m) || !get_mat(root, "distortion_coefficients", r, c, d)) return CVX_ERR_INVALID_ARG;
    if (m.size() < 9) { cvx::set_last_error("camera_matrix needs 9 entries"); return CVX_ERR_INVALID_ARG; }
    if (width) *width = (int)w;
    if (height) *height = (int)h;
    if (K) { K[0] = m[0]; K[1] = m[4]; K[2] = m[2]; K[3] = m[5]; }     // fx, fy, cx, cy (camera.hpp:154-157)
    if (dist) for (int i = 0; i < 5; ++i) dist[i] = i < (int)d.size() ? d[i] : 0.0;   // OpenCV order k1 k2 p1 p2 k3
    return CVX_OK;
}

// PinholeCamera::write (camera.hpp:164-180) with the default 5x1 distortion matrix
extern "C" int cvx_camera_yaml_write(const char *path, int width, int height, const double K[4], const double dist[5]) {
    if (!path || !K || !dist) return CVX_ERR_INVALID_ARG;
    std::ofstream f(path);
    if (!f) { cvx::set_last_error(std::string("cannot write ") + path); return CVX_ERR_INVALID_ARG; }
    const double M[9] = {K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1};
    f << "image_width: " << width << "\nimage_height: " << height << "\ncamera_matrix:\n  rows: 3\n  cols: 3\n  data:\n";
    for (double v : M) f << "    - " << yaml_double(v) << "\n";
    f << "distortion_coefficients:\n  rows: 5\n  cols: 1\n  data:\n";
    for (int i = 0; i < 5; ++i) f << "    - " << yaml_double(dist[i]) << "\n";
    return f ? CVX_OK : CVX_ERR_INVALID_ARG;
}

// CameraRobotConfiguration::readRobotConfiguration (configuration.hpp:44-91).  The pose is
// Translation(x, y, z) * q.normalized().toRotationMatrix() from qx, qy, qz, qw; when those keys are absent but
// roll / pitch / yaw are present (as in the reference's own scripts/configuration.yaml, which its reader
// rejects), the formula the reference keeps commented out is used: Rx(roll) Ry(pitch) Rz(yaw).
extern "C" int cvx_robot_config_read(const char *path, char *base_frame, char *tool_frame, char *camera_frame,
                                     size_t cap, double base_to_tag[3], double initial_cam[16]) {
    if (!path) return CVX_ERR_INVALID_ARG;
    YNode root;
    if (!load_yaml(path, root)) return CVX_ERR_INVALID_ARG;
    for (const char *k : {"base_frame", "tool_frame", "camera_frame"})
        if (!root.map.count(k)) { cvx::set_last_error(std::string("missing key ") + k); return CVX_ERR_INVALID_ARG; }
    set_str(base_frame, cap, root.map["base_frame"].scalar);
    set_str(tool_frame, cap, root.map["tool_frame"].scalar);
    set_str(camera_frame, cap, root.map["camera_frame"].scalar);
    double tag[3];
    if (!get_double(root, "base_to_tag_x", tag[0]) || !get_double(root, "base_to_tag_y", tag[1]) ||
        !get_double(root, "base_to_tag_z", tag[2]))
        return CVX_ERR_INVALID_ARG;
    if (base_to_tag) std::memcpy(base_to_tag, tag, sizeof tag);
    auto it = root.map.find("initial_cam_estimation");
    if (it == root.map.end()) { cvx::set_last_error("missing key initial_cam_estimation"); return CVX_ERR_INVALID_ARG; }
    const YNode &e = it->second;
    double x, y, z;
    if (!get_double(e, "x", x) || !get_double(e, "y", y) || !get_double(e, "z", z)) return CVX_ERR_INVALID_ARG;
    double R[9];    // row-major
    if (e.map.count("qx")) {
        double qx, qy, qz, qw;
        if (!get_double(e, "qx", qx) || !get_double(e, "qy", qy) || !get_double(e, "qz", qz) || !get_double(e, "qw", qw))
            return CVX_ERR_INVALID_ARG;
        const double n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
        const double p[7] = {qw / n, qx / n, qy / n, qz / n, 0, 0, 0};
        double T[16];
        cvx_quat_trans_to_affine(p, T);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[j * 4 + i];
    } else {
        double roll, pitch, yaw;
        if (!get_double(e, "roll", roll) || !get_double(e, "pitch", pitch) || !get_double(e, "yaw", yaw)) {
            cvx::set_last_error("initial_cam_estimation needs qx,qy,qz,qw (or roll,pitch,yaw)");
            return CVX_ERR_INVALID_ARG;
        }
        const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch),
                     cy = std::cos(yaw), sy = std::sin(yaw);
        // Rx(roll) * Ry(pitch) * Rz(yaw)
        const double r[9] = {cp * cy, -cp * sy, sp,
                             sr * sp * cy + cr * sy, -sr * sp * sy + cr * cy, -sr * cp,
                             -cr * sp * cy + sr * sy, cr * sp * sy + sr * cy, cr * cp};
        std::memcpy(R, r, sizeof r);
    }
    if (initial_cam) {
        std::memset(initial_cam, 0, 16 * sizeof(double));
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) initial_cam[j * 4 + i] = R[i * 3 + j];
        initial_cam[12] = x; initial_cam[13] = y; initial_cam[14] = z; initial_cam[15] = 1.0;
    }
    return CVX_OK;
}

// CameraRobotConfiguration::writeCameraTranformationToFile (configuration.hpp:31-43): a ROS launch file with
// one static_transform_publisher; X is the final camera estimate (column-major 4x4)
extern "C" int cvx_robot_config_write_launch(const char *path, const char *tool_frame, const char *camera_frame,
                                             const double X[16]) {
    if (!path || !tool_frame || !camera_frame || !X) return CVX_ERR_INVALID_ARG;
    std::ofstream strm(path);
    if (!strm) { cvx::set_last_error(std::string("cannot write ") + path); return CVX_ERR_INVALID_ARG; }
    double q[4], t[3];
    cvx_affine_to_quat_trans(X, q, t);
    const std::string tool(tool_frame);
    const std::string tail = tool.size() >= 4 ? tool.substr(tool.size() - 4) : tool;
    strm << "<?xml version=\"1.0\"?>" << std::endl;
    strm << "<launch>" << std::endl;
    strm << "<node name=\"" << tail << "_to_camera\" pkg=\"tf\" type=\"static_transform_publisher\" args=\"";
    strm << t[0] << " " << t[1] << " " << t[2];
    strm << " " << q[1] << " " << q[2] << " " << q[3] << " " << q[0];
    strm << " " << tool << " " << camera_frame << " 100\"/>" << std::endl;
    strm << "</launch>" << std::endl;
    return strm ? CVX_OK : CVX_ERR_INVALID_ARG;
}
