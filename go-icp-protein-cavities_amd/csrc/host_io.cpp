// host_io.cpp -- file formats either side of the hot path, restated from their
// behaviour: config.txt (reference ConfigMap.cpp:17-60, jly_main.cpp:231-270), cavity mol2 -> centred/scaled
// -> 6-significant-digit .xyz text -> float (transformation.cpp:282-361, jly_main.cpp:67-99,272-314),
// .cfpfh rows, result files (jly_main.cpp:130-141, transformation.cpp:403-417), protein transform + RMSD
// (transformation.cpp:423-539).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "goicp.hpp"
#include "host_io.hpp"

namespace goicp {

// ---- config -----------------------------------------------------------------------------------------
bool read_config_map(const std::string& path, std::map<std::string, std::string>& kv) {
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) return false;
    std::string line;
    while (std::getline(fin, line)) {
        if (!line.empty() && line[0] == '#') continue;
        if (!line.empty() && line[line.size() - 1] == '\r') line.resize(line.size() - 1);
        std::vector<std::string> tok;
        std::string cur;
        for (char ch : line) {
            if (ch == ' ' || ch == '=' || ch == ';') { if (!cur.empty()) { tok.push_back(cur); cur.clear(); } }
            else cur.push_back(ch);
        }
        if (!cur.empty()) tok.push_back(cur);
        if (tok.size() == 2) kv[tok[0]] = tok[1];
    }
    return true;
}

const char* const kConfigKeys[18] = {"MSEThresh", "rotMinX", "rotMinY", "rotMinZ", "rotWidth", "transMinX", "transMinY", "transMinZ",
                                     "transWidth", "trimFraction", "regularization", "regularizationNeighbors", "regularizationFPFH",
                                     "cfpfh", "norm", "ponderation", "distTransSize", "distTransExpandFactor"};

void apply_config(GoICP& g, const double* v) {     // field by field as readConfig, jly_main.cpp:236-264
    g.MSEThresh = (float)v[0];
    g.initNodeRot.a = (float)v[1]; g.initNodeRot.b = (float)v[2]; g.initNodeRot.c = (float)v[3]; g.initNodeRot.w = (float)v[4];
    g.initNodeTrans.x = (float)v[5]; g.initNodeTrans.y = (float)v[6]; g.initNodeTrans.z = (float)v[7]; g.initNodeTrans.w = (float)v[8];
    g.trimFraction = (float)v[9];
    g.regularization = (float)v[10]; g.regularizationNeighbors = (float)v[11]; g.regularizationFPFH = (float)v[12];
    g.cfpfh = (int)v[13]; g.norm = (int)v[14]; g.ponderation = (int)v[15];
    g.doTrim = true;
    if (g.trimFraction < 0.001) g.doTrim = false;
    g.dt.SIZE = (int)v[16]; g.dt.expandFactor = v[17];
}

bool read_config(const std::string& path, double* v) {
    std::map<std::string, std::string> kv;
    if (!read_config_map(path, kv)) return false;
    for (int k = 0; k < 18; k++) {
        auto it = kv.find(kConfigKeys[k]);
        const std::string s = it == kv.end() ? std::string() : it->second;
        // getF = atof (double), getI = atoi (ConfigMap.cpp:88-110); cfpfh/norm/ponderation/distTransSize are ints
        if (k == 13 || k == 14 || k == 15 || k == 16) v[k] = (double)atoi(s.c_str());
        else v[k] = atof(s.c_str());
    }
    return true;
}

// ---- mol2 ---------------------------------------------------------------------------------------------
bool read_mol2(const std::string& path, std::vector<Atom>& out) {
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    out.clear();
    std::string line;
    bool atoms = false;
    while (std::getline(f, line)) {
        if (!atoms) { if (line.find("@<TRIPOS>ATOM") != std::string::npos) atoms = true; continue; }
        if (line.compare(0, 9, "@<TRIPOS>") == 0) break;
        std::istringstream ss(line);
        std::string id, name; Atom a;
        if (!(ss >> id >> name >> a.x >> a.y >> a.z)) break;
        a.name = name; a.c = string_to_prop(name);
        out.push_back(a);
    }
    return true;
}

double normalise(std::vector<Atom>& cloud, double mean[3]) {       // transformation.cpp:311-335
    const size_t n = cloud.size();
    mean[0] = mean[1] = mean[2] = 0;
    for (const Atom& p : cloud) { mean[0] += p.x; mean[1] += p.y; mean[2] += p.z; }
    mean[0] /= n; mean[1] /= n; mean[2] /= n;
    double maxNorm = 0;
    for (Atom& p : cloud) {
        p.x -= mean[0]; p.y -= mean[1]; p.z -= mean[2];
        const double nr = std::sqrt(std::pow(p.x, 2) + std::pow(p.y, 2) + std::pow(p.z, 2));
        if (nr > maxNorm) maxNorm = nr;
    }
    return maxNorm;
}

bool write_xyzc(const std::string& path, const std::vector<Atom>& cloud) {   // transformation.cpp:340-350 (default 6 significant digits)
    std::ofstream f(path.c_str());
    if (!f.is_open()) return false;
    f << cloud.size() << std::endl;
    for (const Atom& p : cloud) f << p.x << " " << p.y << " " << p.z << " " << p.c << std::endl;
    return true;
}

// loadPointCloud, jly_main.cpp:272-314: "N" then "x y z c" rows, plus 41 floats per point from the .cfpfh file
bool load_xyzc(const std::string& path, const std::string& cfpfh_path, std::vector<POINT3D>& pts) {
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    int n = 0; f >> n;
    pts.assign((size_t)(n > 0 ? n : 0), POINT3D());
    std::ifstream cf;
    if (!cfpfh_path.empty()) { cf.open(cfpfh_path.c_str()); if (!cf.is_open()) return false; }
    for (int i = 0; i < n; i++) {
        f >> pts[i].x >> pts[i].y >> pts[i].z >> pts[i].c;
        pts[i].cfpfh.assign(41, 0.f);
        if (cf.is_open()) for (int j = 0; j < 41; j++) { float b = 0; cf >> b; pts[i].cfpfh[j] = b; }
    }
    return true;
}

bool load_txt(const std::string& path, std::vector<POINT3D>& pts) {   // upstream Go-ICP demo format
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    int n = 0; f >> n;
    pts.assign((size_t)(n > 0 ? n : 0), POINT3D());
    for (int i = 0; i < n; i++) { f >> pts[i].x >> pts[i].y >> pts[i].z; pts[i].c = 0; pts[i].cfpfh.assign(41, 0.f); }
    return true;
}

// ---- outputs ------------------------------------------------------------------------------------------
bool write_output(const std::string& path, double seconds, const GoICP& g) {   // jly_main.cpp:130-141
    std::ofstream o(path.c_str());
    if (!o.is_open()) return false;
    o << "Time: " << seconds << std::endl;
    o << "Rotation Matrix: " << std::endl << g.optR << std::endl;
    o << "Translation Vector: " << std::endl << g.optT << std::endl;
    o << "Error: " << g.optError << std::endl;
    o << "Compatibilities: " << g.Nd - g.optComp << std::endl;
    return true;
}

void rescaled_translation(const GoICP& g, double scale, const double src_mean[3], const double tgt_mean[3], double out[3]) {
    // transformation.cpp:410-412
    for (int k = 0; k < 3; k++)
        out[k] = -(g.optR.val[k][0] * src_mean[0] + g.optR.val[k][1] * src_mean[1] + g.optR.val[k][2] * src_mean[2]) + (scale * g.optT.val[k][0]) + tgt_mean[k];
}

bool write_rescaled(const std::string& path, double seconds, const GoICP& g, const double t[3]) {   // transformation.cpp:403-417
    std::ofstream f(path.c_str());
    if (!f.is_open()) return false;
    f << "Time: " << seconds << std::endl;
    f << "Rotation Matrix:" << std::endl << "   " << g.optR.val[0][0] << "   " << g.optR.val[0][1] << "   " << g.optR.val[0][2] << std::endl;
    f << "   " << g.optR.val[1][0] << "   " << g.optR.val[1][1] << "   " << g.optR.val[1][2] << std::endl;
    f << "   " << g.optR.val[2][0] << "   " << g.optR.val[2][1] << "   " << g.optR.val[2][2] << std::endl;
    f << "Translation Vector:" << std::endl << "   " << t[0] << std::endl << "   " << t[1] << std::endl << "   " << t[2] << std::endl;
    f << "Error: " << (double)g.optError << std::endl;
    return true;
}

// readOutput-compatible parse of "<output>.txt" / "_rescaled.txt": Time, 3x3, 3x1, Error
bool read_output(const std::string& path, double R[9], double t[3], double* seconds, double* error) {
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    std::vector<double> nums;
    std::string tok;
    while (f >> tok) {
        char* end = nullptr;
        const double v = strtod(tok.c_str(), &end);
        if (end != tok.c_str() && *end == 0) nums.push_back(v);
    }
    if (nums.size() < 14) return false;
    if (seconds) *seconds = nums[0];
    for (int k = 0; k < 9; k++) R[k] = nums[1 + k];
    for (int k = 0; k < 3; k++) t[k] = nums[10 + k];
    if (error) *error = nums[13];
    return true;
}

// ---- protein transform + RMSD ----------------------------------------------------
static bool backbone_block(const std::string& path, std::vector<Atom>& out) {   // getAtomBlock, transformation.cpp:423-448
    std::vector<Atom> all;
    if (!read_mol2(path, all)) return false;
    out.clear();
    for (const Atom& a : all) if (a.c == C || a.c == CA || a.c == N || a.c == O) out.push_back(a);
    return true;
}

float compute_rmsd(const std::string& aligned, const std::string& rot) {         // transformation.cpp:453-464
    std::vector<Atom> A, B;
    if (!backbone_block(aligned, A) || !backbone_block(rot, B)) return NAN;
    float rmsd = 0;
    for (size_t i = 0; i < A.size() && i < B.size(); i++)
        rmsd += std::pow((A[i].x - B[i].x), 2) + std::pow((A[i].y - B[i].y), 2) + std::pow((A[i].z - B[i].z), 2);
    rmsd = std::sqrt(rmsd / A.size());
    return rmsd;
}

bool apply_transformation_protein(const std::string& output_txt, const std::string& protein, const std::string& out_path) {
    // transformation.cpp:469-539: coordinates replaced by to_string(R*p+t) (6 decimals), other columns kept, tab separated
    double R[9], t[3];
    if (!read_output(output_txt, R, t, nullptr, nullptr)) return false;
    std::ifstream in(protein.c_str());
    std::ofstream out(out_path.c_str());
    if (!in.is_open() || !out.is_open()) return false;
    std::string line;
    int state = 0;   // 0 header, 1 atoms, 2 tail
    while (std::getline(in, line)) {
        if (state == 0) { out << line << std::endl; if (line == "@<TRIPOS>ATOM") state = 1; continue; }
        if (state == 2) { out << line << std::endl; continue; }
        std::istringstream ss(line);
        std::string s[9];
        int n = 0;
        while (n < 9 && (ss >> s[n])) n++;
        // the reference prints the tag, then its line-oriented copy loop emits the (empty) rest of that line (transformation.cpp:505-520)
        if (n >= 1 && s[0] == "@<TRIPOS>BOND") { out << s[0] << std::endl << std::endl; state = 2; continue; }
        if (n < 9) { out << line << std::endl; continue; }
        const double x = atof(s[2].c_str()), y = atof(s[3].c_str()), z = atof(s[4].c_str());
        const double nx = R[0] * x + R[1] * y + R[2] * z + t[0], ny = R[3] * x + R[4] * y + R[5] * z + t[1], nz = R[6] * x + R[7] * y + R[8] * z + t[2];
        out << s[0] << "\t" << s[1] << "\t" << std::to_string(nx) << "\t" << std::to_string(ny) << "\t" << std::to_string(nz) << "\t" << s[5] << "\t" << s[6]
            << "\t" << s[7] << "\t" << s[8] << std::endl;
    }
    return true;
}

}  // namespace goicp
