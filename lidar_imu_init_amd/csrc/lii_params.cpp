// libliinit_hip — the reference's parameter surface (config/*.yaml + launch/*.launch) as a host-side loader.
// The reference reads its parameters from the ROS parameter server (nh.param<> block, src/laserMapping.cpp:767-799), which
// roslaunch fills from `<rosparam command="load" file=".../config/X.yaml"/>` and `<param name=... value=.../>` tags
// (launch/*.launch:6-12).  A host that links this library without ROS gets the same names, defaults and override order
// from lii_params_defaults / lii_params_load_yaml / lii_params_load_launch, and lii_params_apply turns them into the
// structs of the C-ABI.  Only the YAML subset those files use is understood: nested block maps, scalars, quoted strings,
// flow sequences of numbers, '#' comments.  Pure host code (no device work).
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/liinit_hip.h"

namespace {

thread_local std::string g_perr;
int pfail(int code, const std::string& m) {
  g_perr = m;
  return code;
}

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) a++;
  while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}
// cuts a '#' comment that is not inside quotes
std::string strip_comment(const std::string& s) {
  char q = 0;
  for (size_t i = 0; i < s.size(); i++) {
    const char c = s[i];
    if (q) { if (c == q) q = 0; }
    else if (c == '"' || c == '\'') q = c;
    else if (c == '#' && (i == 0 || std::isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
  }
  return s;
}
std::string unquote(const std::string& s) {
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
  return s;
}

typedef std::map<std::string, std::string> Flat;  // "section/key" -> raw scalar text (flow sequences keep their brackets)

// block maps by indentation -> flat "a/b/c" names (the names nh.param uses)
bool parse_yaml(std::istream& in, Flat& out, std::string& why) {
  std::vector<std::pair<int, std::string>> stack;  // (indent, key) of the open maps
  std::string line;
  int lineno = 0;
  while (std::getline(in, line)) {
    lineno++;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::string body = strip_comment(line);
    if (trim(body).empty()) continue;
    if (body.find('\t') != std::string::npos && trim(body.substr(0, body.find_first_not_of(" \t"))).empty()) {
      for (char& c : body) if (c == '\t') c = ' ';  // tolerate tabs (YAML forbids them; roslaunch's loader does not see them here)
    }
    const int indent = (int)body.find_first_not_of(' ');
    const std::string t = trim(body);
    if (t == "---" || t == "...") continue;
    const size_t colon = t.find(':');
    if (colon == std::string::npos) { why = "line " + std::to_string(lineno) + ": expected `key: value`"; return false; }
    const std::string key = unquote(trim(t.substr(0, colon)));
    const std::string val = trim(t.substr(colon + 1));
    while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
    std::string full;
    for (auto& s : stack) full += s.second + "/";
    full += key;
    if (val.empty()) stack.emplace_back(indent, key);
    else out[full] = val;
  }
  return true;
}

bool to_double(const std::string& s, double& v) {
  const std::string u = unquote(s);
  if (u.empty()) return false;
  errno = 0;
  char* end = nullptr;
  v = std::strtod(u.c_str(), &end);
  return errno == 0 && end && *end == 0;
}
bool to_bool(const std::string& s, int32_t& v) {
  std::string u = unquote(s);
  for (char& c : u) c = (char)std::tolower((unsigned char)c);
  if (u == "true" || u == "yes" || u == "on" || u == "1") { v = 1; return true; }
  if (u == "false" || u == "no" || u == "off" || u == "0") { v = 0; return true; }
  double d;
  if (to_double(u, d)) { v = d != 0; return true; }
  return false;
}
bool to_list(const std::string& s, std::vector<double>& v) {
  const std::string u = trim(s);
  if (u.size() < 2 || u.front() != '[' || u.back() != ']') return false;
  std::stringstream ss(u.substr(1, u.size() - 2));
  std::string item;
  v.clear();
  while (std::getline(ss, item, ',')) {
    double d;
    if (trim(item).empty()) continue;
    if (!to_double(trim(item), d)) return false;
    v.push_back(d);
  }
  return true;
}
void set_str(char* dst, size_t cap, const std::string& s) {
  std::snprintf(dst, cap, "%s", unquote(s).c_str());
}

// name -> field (the nh.param names of src/laserMapping.cpp:770-799); unknown names are ignored as the node would
bool assign(lii_params* p, const std::string& name, const std::string& raw, std::string& why) {
  double d = 0;
  int32_t b = 0;
  auto num = [&](double& dst) { if (!to_double(raw, d)) { why = name + ": not a number: " + raw; return false; } dst = d; return true; };
  auto inum = [&](int32_t& dst) { if (!to_double(raw, d)) { why = name + ": not a number: " + raw; return false; } dst = (int32_t)d; return true; };
  auto flag = [&](int32_t& dst) { if (!to_bool(raw, b)) { why = name + ": not a bool: " + raw; return false; } dst = b; return true; };
  auto vec3 = [&](double* dst, int32_t& n) {
    std::vector<double> v;
    if (!to_list(raw, v)) { why = name + ": not a list: " + raw; return false; }
    n = (int32_t)std::min<size_t>(v.size(), 3);
    for (int i = 0; i < n; i++) dst[i] = v[i];
    return true;
  };
  if (name == "max_iteration") return inum(p->max_iteration);
  if (name == "point_filter_num") return inum(p->point_filter_num);
  if (name == "map_file_path") { set_str(p->map_file_path, sizeof(p->map_file_path), raw); return true; }
  if (name == "common/lid_topic") { set_str(p->lid_topic, sizeof(p->lid_topic), raw); return true; }
  if (name == "common/imu_topic") { set_str(p->imu_topic, sizeof(p->imu_topic), raw); return true; }
  if (name == "mapping/filter_size_surf") return num(p->filter_size_surf);
  if (name == "mapping/filter_size_map") return num(p->filter_size_map);
  if (name == "cube_side_length") return num(p->cube_side_length);
  if (name == "mapping/det_range") return num(p->det_range);
  if (name == "mapping/gyr_cov") return num(p->gyr_cov);
  if (name == "mapping/acc_cov") return num(p->acc_cov);
  if (name == "mapping/grav_cov") return num(p->grav_cov);
  if (name == "mapping/b_gyr_cov") return num(p->b_gyr_cov);
  if (name == "mapping/b_acc_cov") return num(p->b_acc_cov);
  if (name == "preprocess/blind") return num(p->blind);
  if (name == "preprocess/lidar_type") return inum(p->lidar_type);
  if (name == "preprocess/scan_line") return inum(p->scan_line);
  if (name == "preprocess/feature_extract_en") return flag(p->feature_extract_en);
  if (name == "initialization/cut_frame") return flag(p->cut_frame);
  if (name == "initialization/cut_frame_num") return inum(p->cut_frame_num);
  if (name == "initialization/orig_odom_freq") return inum(p->orig_odom_freq);
  if (name == "initialization/online_refine_time") return num(p->online_refine_time);
  if (name == "initialization/mean_acc_norm") return num(p->mean_acc_norm);
  if (name == "initialization/data_accum_length") return num(p->data_accum_length);
  if (name == "initialization/Rot_LI_cov") return vec3(p->Rot_LI_cov, p->n_Rot_LI_cov);
  if (name == "initialization/Trans_LI_cov") return vec3(p->Trans_LI_cov, p->n_Trans_LI_cov);
  if (name == "publish/path_en") return flag(p->path_en);
  if (name == "publish/scan_publish_en") return flag(p->scan_publish_en);
  if (name == "publish/dense_publish_en") return flag(p->dense_publish_en);
  if (name == "publish/scan_bodyframe_pub_en") return flag(p->scan_bodyframe_pub_en);
  if (name == "runtime_pos_log_enable") return flag(p->runtime_pos_log_enable);
  if (name == "pcd_save/pcd_save_en") return flag(p->pcd_save_en);
  if (name == "pcd_save/interval") return inum(p->pcd_save_interval);
  return true;  // a name the node never asks for
}

// attribute value of the first `attr="..."` inside a tag's text
bool xml_attr(const std::string& tag, const std::string& attr, std::string& out) {
  size_t pos = 0;
  while ((pos = tag.find(attr, pos)) != std::string::npos) {
    const bool word_start = pos == 0 || std::isspace((unsigned char)tag[pos - 1]);
    size_t q = pos + attr.size();
    while (q < tag.size() && std::isspace((unsigned char)tag[q])) q++;
    if (word_start && q < tag.size() && tag[q] == '=') {
      q++;
      while (q < tag.size() && std::isspace((unsigned char)tag[q])) q++;
      if (q < tag.size() && (tag[q] == '"' || tag[q] == '\'')) {
        const char quote = tag[q];
        const size_t e = tag.find(quote, q + 1);
        if (e == std::string::npos) return false;
        out = tag.substr(q + 1, e - q - 1);
        return true;
      }
    }
    pos += attr.size();
  }
  return false;
}
std::string dir_of(const std::string& path) {
  const size_t s = path.find_last_of('/');
  return s == std::string::npos ? std::string(".") : path.substr(0, s);
}
bool file_exists(const std::string& p) {
  std::ifstream f(p);
  return f.good();
}

}  // namespace

extern "C" {

const char* lii_params_last_error(void) { return g_perr.c_str(); }

int lii_params_defaults(lii_params* p) {
  if (!p) return pfail(LII_ERR_INVALID, "lii_params_defaults: NULL");
  std::memset(p, 0, sizeof(*p));
  p->struct_size = sizeof(lii_params);
  // the third argument of every nh.param<> call, src/laserMapping.cpp:770-799
  p->max_iteration = 4;
  p->point_filter_num = 2;
  std::snprintf(p->lid_topic, sizeof(p->lid_topic), "/livox/lidar");
  std::snprintf(p->imu_topic, sizeof(p->imu_topic), "/livox/imu");
  p->filter_size_surf = 0.5;
  p->filter_size_map = 0.5;
  p->cube_side_length = 200;
  p->det_range = 300.0;
  p->gyr_cov = 0.1; p->acc_cov = 0.1; p->grav_cov = 0.001; p->b_gyr_cov = 0.0001; p->b_acc_cov = 0.0001;
  p->blind = 1.0;
  p->lidar_type = LII_LIDAR_AVIA;
  p->scan_line = 16;
  p->feature_extract_en = 0;
  p->cut_frame = 1;
  p->cut_frame_num = 1;
  p->orig_odom_freq = 10;
  p->online_refine_time = 20.0;
  p->mean_acc_norm = 9.81;
  p->data_accum_length = 300;
  p->path_en = 1; p->scan_publish_en = 1; p->dense_publish_en = 1; p->scan_bodyframe_pub_en = 1;
  p->runtime_pos_log_enable = 0;
  p->pcd_save_en = 0;
  p->pcd_save_interval = -1;
  return LII_OK;
}

int lii_params_set(lii_params* p, const char* name, const char* value) {
  if (!p || !name || !value || p->struct_size != sizeof(lii_params)) return pfail(LII_ERR_INVALID, "lii_params_set: bad arguments");
  std::string why;
  std::string n = name;
  while (!n.empty() && n.front() == '/') n.erase(n.begin());  // "/max_iteration" and "max_iteration" name the same global parameter
  if (!assign(p, n, value, why)) return pfail(LII_ERR_INVALID, why);
  return LII_OK;
}

int lii_params_load_yaml(const char* path, lii_params* p) {
  if (!path || !p || p->struct_size != sizeof(lii_params)) return pfail(LII_ERR_INVALID, "lii_params_load_yaml: bad arguments");
  std::ifstream f(path);
  if (!f.good()) return pfail(LII_ERR_INVALID, std::string("cannot open ") + path);
  Flat flat;
  std::string why;
  if (!parse_yaml(f, flat, why)) return pfail(LII_ERR_INVALID, std::string(path) + ": " + why);
  for (auto& kv : flat)
    if (!assign(p, kv.first, kv.second, why)) return pfail(LII_ERR_INVALID, std::string(path) + ": " + why);
  return LII_OK;
}

int lii_params_load_launch(const char* launch_path, const char* config_dir, lii_params* p) {
  if (!launch_path || !p || p->struct_size != sizeof(lii_params)) return pfail(LII_ERR_INVALID, "lii_params_load_launch: bad arguments");
  std::ifstream f(launch_path);
  if (!f.good()) return pfail(LII_ERR_INVALID, std::string("cannot open ") + launch_path);
  std::stringstream ss;
  ss << f.rdbuf();
  std::string xml = ss.str();
  for (size_t a; (a = xml.find("<!--")) != std::string::npos;) {  // comments out
    const size_t b = xml.find("-->", a + 4);
    xml.erase(a, b == std::string::npos ? std::string::npos : b + 3 - a);
  }
  // two passes like roslaunch: every <rosparam load> first, then the <param> tags (distinct names in the shipped files)
  std::vector<std::pair<std::string, std::string>> overrides;
  size_t pos = 0;
  while ((pos = xml.find('<', pos)) != std::string::npos) {
    const size_t end = xml.find('>', pos);
    if (end == std::string::npos) break;
    const std::string tag = xml.substr(pos + 1, end - pos - 1);
    pos = end + 1;
    std::string name = tag.substr(0, tag.find_first_of(" \t\r\n/"));
    if (name == "rosparam") {
      std::string cmd, file;
      if (xml_attr(tag, "command", cmd) && cmd == "load" && xml_attr(tag, "file", file)) {
        // $(find pkg)/config/X.yaml -> <config_dir>/X.yaml, or <launch dir>/../config/X.yaml
        std::string rel = file;
        const size_t fe = rel.find("$(find");
        if (fe != std::string::npos) {
          const size_t close = rel.find(')', fe);
          rel = close == std::string::npos ? rel : rel.substr(close + 1);
        }
        while (!rel.empty() && rel.front() == '/') rel.erase(rel.begin());
        const std::string base = rel.substr(rel.find_last_of('/') == std::string::npos ? 0 : rel.find_last_of('/') + 1);
        std::vector<std::string> tries;
        if (config_dir && *config_dir) tries.push_back(std::string(config_dir) + "/" + base);
        tries.push_back(dir_of(launch_path) + "/../" + rel);
        tries.push_back(dir_of(launch_path) + "/" + base);
        if (file.find("$(") == std::string::npos) tries.push_back(file);
        std::string found;
        for (auto& t : tries)
          if (file_exists(t)) { found = t; break; }
        if (found.empty()) return pfail(LII_ERR_INVALID, std::string(launch_path) + ": cannot resolve rosparam file " + file);
        const int rc = lii_params_load_yaml(found.c_str(), p);
        if (rc != LII_OK) return rc;
      }
    } else if (name == "param") {
      std::string pn, pv;
      if (xml_attr(tag, "name", pn) && xml_attr(tag, "value", pv)) overrides.emplace_back(pn, pv);
    }
  }
  for (auto& o : overrides) {
    const int rc = lii_params_set(p, o.first.c_str(), o.second.c_str());
    if (rc != LII_OK) return pfail(rc, std::string(launch_path) + ": " + g_perr);
  }
  return LII_OK;
}

int lii_params_apply(const lii_params* p, int32_t device, int32_t max_scan_points, int32_t max_map_points, lii_config* cfg,
                     lii_ingest_opts* ingest, lii_iekf_opts* opts, float* leaf) {
  if (!p || p->struct_size != sizeof(lii_params)) return pfail(LII_ERR_INVALID, "lii_params_apply: bad lii_params");
  if (!(p->filter_size_map > 0) || !(p->filter_size_surf > 0) || p->max_iteration < 1 || p->cut_frame_num < 1 || p->cut_frame_num > 64 ||
      p->point_filter_num < 1 || p->lidar_type < LII_LIDAR_AVIA || p->lidar_type > LII_LIDAR_ROBOSENSE || p->scan_line < 1)
    return pfail(LII_ERR_INVALID, "lii_params_apply: parameter out of range (filter sizes, max_iteration, cut_frame_num, point_filter_num, lidar_type, scan_line)");
  if (cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(lii_config);
    cfg->device = device;
    cfg->max_scan_points = max_scan_points;
    cfg->max_map_points = max_map_points;
    cfg->map_cell_size = 0.f;                                  // derived from the down-sample box
    cfg->map_downsample_size = (float)p->filter_size_map;      // ikdtree.set_downsample_param(filter_size_map_min), laserMapping.cpp:922
    cfg->max_match_dist2 = 5.0f;                               // Nearest_Search(..., 5), :980
    cfg->plane_threshold = 0.1;                                // esti_plane(pabcd, points_near, 0.1), :997
    cfg->laser_point_cov_inv = 1000.0;                         // 1 / LASER_POINT_COV, :59
  }
  if (ingest) {
    std::memset(ingest, 0, sizeof(*ingest));
    ingest->struct_size = sizeof(lii_ingest_opts);
    ingest->lidar_type = p->lidar_type;
    ingest->n_scans = p->scan_line;
    ingest->point_filter_num = p->point_filter_num;
    ingest->blind = p->blind;
    ingest->cut_frame_num = p->cut_frame ? p->cut_frame_num : 0;  // (0: the callbacks' non-cutting branch, Preprocess::process - laserMapping.cpp:337-342, :374-379)
    ingest->scan_count = 0;  // per message: the caller's running count
    ingest->stamp_s = 0.0;   // per message
  }
  if (opts) {
    opts->max_iterations = p->max_iteration;
    opts->imu_en = 0;  // LiDAR-only odometry until LI_Initialization has run (imu_en starts false, laserMapping.cpp:93)
  }
  if (leaf) *leaf = (float)p->filter_size_surf;
  return LII_OK;
}

}  // extern "C"
