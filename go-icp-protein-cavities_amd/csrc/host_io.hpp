// host_io.hpp -- file formats around the hot path (see host_io.cpp for the reference lines restated).
#ifndef GOICP_HOST_IO_HPP
#define GOICP_HOST_IO_HPP
#include <map>
#include <string>
#include <vector>
#include "goicp.hpp"

namespace goicp {

struct Atom { double x = 0, y = 0, z = 0; int c = 0; std::string name; };

extern const char* const kConfigKeys[18];
bool read_config_map(const std::string& path, std::map<std::string, std::string>& kv);
bool read_config(const std::string& path, double* cfg18);
void apply_config(GoICP& g, const double* cfg18);

bool read_mol2(const std::string& path, std::vector<Atom>& out);
double normalise(std::vector<Atom>& cloud, double mean[3]);
bool write_xyzc(const std::string& path, const std::vector<Atom>& cloud);
bool load_xyzc(const std::string& path, const std::string& cfpfh_path, std::vector<POINT3D>& pts);
bool load_txt(const std::string& path, std::vector<POINT3D>& pts);

bool write_output(const std::string& path, double seconds, const GoICP& g);
void rescaled_translation(const GoICP& g, double scale, const double src_mean[3], const double tgt_mean[3], double out[3]);
bool write_rescaled(const std::string& path, double seconds, const GoICP& g, const double t[3]);
bool read_output(const std::string& path, double R[9], double t[3], double* seconds, double* error);
float compute_rmsd(const std::string& aligned, const std::string& rot);
bool apply_transformation_protein(const std::string& output_txt, const std::string& protein, const std::string& out_path);

}  // namespace goicp
#endif
