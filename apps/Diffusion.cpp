// Diffusion -- inputs-file front end of the MI355X flow-through tortuosity solver.
//
// Keeps the user-facing boundary of the reference's `Diffusion` application
// (src/props/Diffusion.cpp): `Diffusion <inputs> [key=value ...]`, the same keys
// (:200-223, :605-611, :633; TortuosityHypre.cpp:147-151,254-256), reader dispatch by
// file extension (:262-300), the per-direction loop (:655-706), the console lines and the
// results file (:718-728).  The solve behind it is the HIP library (C ABI), reached
// through the TortuosityMI355 mirror of OpenImpala::TortuosityHypre.
//
// Scope: calculation_method = flow_through (the hot path) and homogenization (first cut of
// the "next" row: three periodic cell problems + D_eff tensor, one GPU or a ring of z-slabs), the REV
// sampling study (rev.do_study), the chi_k / potential plotfiles and the CathodeWrite exporters (cathode.*).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <random>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <string>
#include <vector>

#include "../include/TortuosityMI355.H"
#include "Inputs.H"
#include "CathodeWrite.H"
#include "Plotfile.H"

namespace {

std::string lower(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return std::tolower(c); });
  return s;
}
std::string upper(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return std::toupper(c); });
  return s;
}

// stringToSolverType (Diffusion.cpp:45-58): unknown names are fatal
OpenImpala::TortuosityMI355::SolverType solver_from_string(const std::string& s) {
  using T = OpenImpala::TortuosityMI355::SolverType;
  const std::string l = lower(s);
  if (l == "jacobi") return T::Jacobi;
  if (l == "gmres") return T::GMRES;
  if (l == "flexgmres") return T::FlexGMRES;
  if (l == "pcg") return T::PCG;
  if (l == "bicgstab") return T::BiCGSTAB;
  if (l == "smg") return T::SMG;
  if (l == "pfmg") return T::PFMG;
  throw std::runtime_error("Invalid solver string: '" + s + "'.");
}

// the phase bytes stay in the reader's own buffer: page-locked when a device is present (oi_read_set_pinned), so that the
// upload of the volume -- or of one z-slab per GPU -- is a single DMA transfer instead of a staged pageable copy
struct PhaseBytes {
  std::unique_ptr<std::uint8_t, void (*)(void*)> buf{nullptr, oi_free_host};
  std::size_t n = 0;
  const std::uint8_t* data() const { return buf.get(); }
  std::size_t size() const { return n; }
  const std::uint8_t& operator[](std::size_t k) const { return buf.get()[k]; }
  const std::uint8_t* begin() const { return buf.get(); }
  const std::uint8_t* end() const { return buf.get() + n; }
};
struct Volume {
  PhaseBytes phase;
  int nx = 0, ny = 0, nz = 0;
};

Volume load_volume(const std::filesystem::path& path, double threshold, const oi::Inputs& in) {
  if (!path.has_extension()) throw std::runtime_error("File has no extension: " + path.string());
  const std::string ext = lower(path.extension().string());
  Volume v;
  std::uint8_t* data = nullptr;
  oi_status st = OI_OK;
  if (ext == ".tif" || ext == ".tiff") {
    int bps = 0;
    st = oi_read_tiff(path.c_str(), threshold, &data, &v.nx, &v.ny, &v.nz, &bps);
  } else if (ext == ".dat") {
    st = oi_read_dat(path.c_str(), threshold, &data, &v.nx, &v.ny, &v.nz);
  } else if (ext == ".raw") {
    // the reference's Diffusion does not wire RawReader (Diffusion.cpp:12); tRawReader's keys are used here
    std::string type = "UINT8";
    if (!in.query("raw.width", v.nx) || !in.query("raw.height", v.ny) || !in.query("raw.depth", v.nz))
      throw std::runtime_error("a .raw input needs raw.width, raw.height and raw.depth");
    in.query("raw.data_type", type);
    static const std::map<std::string, int> types = {
        {"UINT8", OI_RAW_UINT8}, {"INT8", OI_RAW_INT8}, {"INT16_LE", OI_RAW_INT16_LE}, {"INT16_BE", OI_RAW_INT16_BE},
        {"UINT16_LE", OI_RAW_UINT16_LE}, {"UINT16_BE", OI_RAW_UINT16_BE}, {"INT32_LE", OI_RAW_INT32_LE},
        {"INT32_BE", OI_RAW_INT32_BE}, {"UINT32_LE", OI_RAW_UINT32_LE}, {"UINT32_BE", OI_RAW_UINT32_BE},
        {"FLOAT32_LE", OI_RAW_FLOAT32_LE}, {"FLOAT32_BE", OI_RAW_FLOAT32_BE}, {"FLOAT64_LE", OI_RAW_FLOAT64_LE},
        {"FLOAT64_BE", OI_RAW_FLOAT64_BE}};
    const auto it = types.find(upper(type));
    if (it == types.end()) throw std::runtime_error("unknown raw.data_type: " + type);
    st = oi_read_raw(path.c_str(), v.nx, v.ny, v.nz, it->second, threshold, &data);
  } else if (ext == ".h5" || ext == ".hdf5") {
    std::string ds = "image";  // Diffusion.cpp:181,205
    in.query("hdf5_dataset", ds);
    st = oi_read_hdf5(path.c_str(), ds.c_str(), threshold, &data, &v.nx, &v.ny, &v.nz);
  } else {
    throw std::runtime_error("Unsupported file extension for full domain load: " + ext);  // Diffusion.cpp:298-300
  }
  if (st != OI_OK) throw std::runtime_error(std::string(oi_status_string(st)) + ": " + oi_last_error());
  v.phase.buf.reset(data);
  v.phase.n = static_cast<std::size_t>(v.nx) * v.ny * v.nz;
  return v;
}

}  // namespace

int main(int argc, char* argv[]) {
  const auto t_start = std::chrono::steady_clock::now();
  if (argc < 2) {
    std::cerr << "usage: Diffusion <inputs file> [key=value ...]\n";
    return 2;
  }
  try {
    oi::Inputs in;
    in.load_file(argv[1]);
    for (int a = 2; a < argc; ++a) in.add_argument(argv[a]);  // command line overrides the file

    // ---- keys of Diffusion.cpp:200-213 (with the stale aliases of build/apps/inputs) ----
    std::string filename, data_path = "./data/", results_path = "./results_diffusion/", hdf5_dataset = "image";
    std::string solver_str = "FlexGMRES", method = "homogenization", output_filename = "results.txt";
    double threshold = 0.5;
    int phase_id = 1, box_size = 32, verbose = 1, write_plotfile = 0;
    in.get("filename", filename);
    in.query("data_path", data_path);
    if (!in.query("results_path", results_path)) in.query("results_dir", results_path);
    in.query("hdf5_dataset", hdf5_dataset);
    if (!in.query("threshold_val", threshold)) in.query("threshold_value", threshold);
    in.query("phase_id", phase_id);
    in.query("solver_type", solver_str);
    in.query("box_size", box_size);  // AMReX max grid size: only the remspot filter depends on it (it runs per box)
    in.query("verbose", verbose);
    in.query("write_plotfile", write_plotfile);
    // debug.write_active_mask (TortuosityHypre.cpp:542-545: read there, its plotfile body elided in the reference's snapshot): here the
    // percolation mask of every direction is written as raw bytes, x fastest, beside the results file
    int write_active_mask = 0;
    in.query("debug.write_active_mask", write_active_mask);
    in.query("calculation_method", method);
    in.query("output_filename", output_filename);

    const std::filesystem::path results_dir(results_path);
    if (!std::filesystem::exists(results_dir)) {
      std::filesystem::create_directories(results_dir);
      if (verbose >= 1) std::cout << "Created results directory: " << results_dir.string() << std::endl;
    }
    const std::filesystem::path input_path = std::filesystem::path(data_path) / filename;
    if (verbose >= 1) std::cout << "Reading full domain data from: " << input_path.string() << std::endl;
    Volume vol;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    try {
      const auto t_read = std::chrono::steady_clock::now();
      oi_read_set_pinned(1);  // (pageable memory when no device is visible: the error is reported where a device is needed)
      // this program builds solver object after solver object (one per direction, one per REV sample, as the reference does):
      // the device arrays of a destroyed one are kept for the next instead of going back to the driver, whose clearing of
      // recycled memory can cost seconds per allocation on MI355X (gpu.pool = 0 turns that off)
      // What the pool may hold is bounded (gpu.pool_cap_gb; default a quarter of the card): blocks no later request fits would
      // otherwise pile up over a REV study.  An allocation the driver refuses gives the pool back and is repeated in any case.
      int use_pool = 1;
      double pool_cap_gb = -1.0;
      in.query("gpu.pool", use_pool);
      in.query("gpu.pool_cap_gb", pool_cap_gb);
      if (pool_cap_gb < 0.0) {
        size_t free_b = 0, total_b = 0;
        pool_cap_gb = oi_dev_mem_info(&free_b, &total_b) == OI_OK ? (double)total_b / 4e9 : 64.0;
      }
      oi_dev_pool(use_pool, pool_cap_gb);
      vol = load_volume(input_path, threshold, in);
      if (verbose >= 1)
        std::cout << "  Read + threshold: " << vol.nx << " x " << vol.ny << " x " << vol.nz << " voxels in " << std::defaultfloat
                  << std::setprecision(4) << since(t_read) << " s (incl. the first use of the device when the buffer is page-locked)\n";
    } catch (const std::exception& e) {
      std::cout << "Error loading full domain data: " << e.what() << std::endl;
      std::cerr << "Full domain data loading failed." << std::endl;
      return 1;
    }

    // ---- REV study, Diffusion.cpp:317-502: random cubic sub-volumes, periodic cell problems on each ----
    bool rev_study = false;
    int rev_num_samples = 3, rev_write_plotfiles = 0, rev_verbose = 1;
    std::string rev_sizes = "32 64 96", rev_solver = "FlexGMRES", rev_results_file = "rev_study_Deff.csv";
    in.query("rev.do_study", rev_study);
    in.query("rev.num_samples", rev_num_samples);
    in.query("rev.sizes", rev_sizes);
    in.query("rev.solver_type", rev_solver);
    in.query("rev.results_file", rev_results_file);
    in.query("rev.write_plotfiles", rev_write_plotfiles);
    in.query("rev.verbose", rev_verbose);

    oi_tort_opts hopts;
    oi_tort_default_opts(&hopts);
    if (!in.query("hypre.eps", hopts.eps)) in.query("hypre_eps", hopts.eps);
    in.query("gpu.maxiter", hopts.maxiter);
    in.query("gpu.device", hopts.device);
    {  // gpu.precond = mg (default: one periodic multigrid cycle per PCG iteration, the role of PFMG in the reference) | none
      std::string hp = "mg";
      in.query("gpu.precond", hp);
      const std::string hl = lower(hp);
      if (hl == "mg" || hl == "multigrid") hopts.precond = OI_PRECOND_MG;
      else if (hl == "none" || hl == "jacobi") hopts.precond = OI_PRECOND_NONE;  // (the diagonal is constant: Jacobi = none)
      else throw std::runtime_error("gpu.precond must be mg, jacobi or none (got '" + hp + "')");
    }
    // ---- cathode.*: the reference's CathodeWrite exporters (src/io/CathodeWrite.cpp; no caller in the reference -- wired here behind
    // keys of their own).  cathode.dandeliion_file / cathode.pybamm_file name the files (in the results directory); the solid volume
    // fraction is the analysed phase's unless cathode.volume_fraction_solid gives one ----
    auto write_cathode_files = [&](double vf_phase) {
      std::string f_dl, f_pb;
      in.query("cathode.dandeliion_file", f_dl);
      in.query("cathode.pybamm_file", f_pb);
      if (f_dl.empty() && f_pb.empty()) return;
      OpenImpala::CathodeParams cp;
      cp.volume_fraction_solid = vf_phase;
      in.query("cathode.volume_fraction_solid", cp.volume_fraction_solid);
      in.query("cathode.particle_radius", cp.particle_radius);
      in.query("cathode.conductivity", cp.active_material_conductivity);
      in.query("cathode.max_concentration", cp.max_concentration);
      const OpenImpala::CathodeWrite w(cp);
      if (!f_dl.empty() && !w.writeDandeLiionParameters((results_dir / f_dl).string())) std::cerr << "Warning: DandeLiion parameter file not written\n";
      if (!f_pb.empty() && !w.writePyBammParameters((results_dir / f_pb).string())) std::cerr << "Warning: PyBamm parameter file not written\n";
    };
    const char* axis_names[3] = {"X", "Y", "Z"};
    // gpu.ngpus > 1 for the FULL-domain cell problems: z-slabs on a RING of ranks (periodic z: the first and the last slab are
    // neighbours), one host thread per slab, RCCL over xGMI or -- gpu.transport = local -- all slabs on one GPU (rehearsal)
    int h_ngpus = 1;
    std::string h_transport = "rccl";
    in.query("gpu.ngpus", h_ngpus);
    in.query("gpu.transport", h_transport);
    // three chi_k solves + tensor on one (sub-)volume; false when a solve does not converge (:545-548)
    // chi_k plotfile of the reference (EffectiveDiffusivityHypre.cpp:648-680): <dir>/effdiff_chi_dir<k> with the components
    // {"chi_k", "active_mask_from_solver"} -- the mask of the cell problem is phase == phase_id
    auto write_chi_plotfile = [&](const std::string& dir, int k, const uint8_t* ph, int nx, int ny, int nz, const std::vector<double>& chi, int vb) {
      std::error_code ec;
      std::filesystem::create_directories(dir, ec);
      if (vb > 0) std::cout << "  Writing solution plotfile for chi_k in direction " << k << "..." << std::endl;
      std::vector<double> mask_d((size_t)nx * ny * nz);
      for (size_t t = 0; t < mask_d.size(); ++t) mask_d[t] = ph[t] == (uint8_t)phase_id ? 1.0 : 0.0;
      const std::string full = dir + "/effdiff_chi_dir" + std::to_string(k);
      if (!OpenImpala::writeSingleLevelPlotfile(full, nx, ny, nz, {chi.data(), mask_d.data()}, {"chi_k", "active_mask_from_solver"}, box_size))
        std::cerr << "Warning: could not write plotfile " << full << "\n";
      else if (vb > 0)
        std::cout << "  Plotfile written to " << full << std::endl;
    };
    // (plot_dirs: null, or the three directories the chi_k plotfiles go to)
    auto homogenize_ring = [&](const uint8_t* ph, int nx, int ny, int nz, int vb, const char* what, double D[9],
                               const std::vector<std::string>* plot_dirs) -> bool {
      const int N = h_ngpus;
      if (N > 64 || N > nz) throw std::runtime_error("gpu.ngpus must be between 1 and min(64, nz)");
      if (h_transport != "rccl" && h_transport != "local") throw std::runtime_error("gpu.transport must be rccl or local");
      if (h_transport == "rccl" && oi_device_count() < N)
        throw std::runtime_error("gpu.ngpus = " + std::to_string(N) + " but only " + std::to_string(oi_device_count()) +
                                 " GPU(s) are visible (gpu.transport = local runs the slabs on one GPU)");
      std::vector<int> zb(N + 1, 0);
      {
        const int unit = (nz / N >= 32) ? 16 : 1, units = nz / unit;
        for (int r = 0; r <= N; ++r) zb[r] = (int)((long long)units * r / N) * unit;
        zb[N] = nz;
      }
      std::vector<oi_comm*> comms(N, nullptr);
      unsigned char uid[OI_COMM_ID_BYTES];
      if (h_transport == "local") {
        const oi_status cs = oi_comm_create_local(comms.data(), N, hopts.device);
        if (cs != OI_OK) throw std::runtime_error(std::string(oi_status_string(cs)) + ": " + oi_last_error());
      } else {
        const oi_status cs = oi_comm_unique_id(uid);
        if (cs != OI_OK) throw std::runtime_error(std::string(oi_status_string(cs)) + ": " + oi_last_error());
      }
      std::vector<std::string> errors(N);
      std::vector<int> allconv(N, 1);
      int conv3[3] = {0, 0, 0}, it3[3] = {0, 0, 0}, layout0 = 0;
      double rel3[3] = {0, 0, 0};
      std::vector<double> Ds((size_t)9 * N, 0.0);
      const size_t plane = (size_t)nx * ny;
      std::vector<double> chi_full[3];   // every slab writes its planes of chi_k here when plotfiles are wanted
      if (plot_dirs)
        for (auto& c : chi_full) c.assign(plane * nz, 0.0);
      std::vector<int> got_chi(3 * N, 0);
      auto work = [&](int r) {
        oi_effdiff* eh = nullptr;
        do {
          if (h_transport == "rccl") {
            const oi_status cs = oi_comm_create(&comms[r], uid, N, r, r);
            if (cs != OI_OK) { errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error(); break; }
          }
          oi_tort_opts o = hopts;
          o.verbose = r == 0 ? vb : 0;
          oi_status cs = oi_effdiff_create_slab(&eh, comms[r], ph + plane * zb[r], 0, nx, ny, nz, zb[r], zb[r + 1] - zb[r], phase_id, &o);
          if (cs != OI_OK) { errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error(); break; }
          if (r == 0) { int64_t nel = 0; (void)oi_effdiff_layout(eh, &layout0, &nel); }
          for (int k = 0; k < 3 && allconv[r]; ++k) {  // (the convergence flag comes from all-reduced sums: every rank leaves the loop alike)
            int conv = 0, iters = 0;
            double rel = 0.0;
            cs = oi_effdiff_solve(eh, k, &conv, &iters, &rel);
            if (cs != OI_OK) { errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error(); allconv[r] = 0; break; }
            if (r == 0) { conv3[k] = conv; it3[k] = iters; rel3[k] = rel; }
            if (!conv) allconv[r] = 0;
            else if (plot_dirs) got_chi[3 * r + k] = oi_effdiff_get_chi(eh, k, chi_full[k].data() + plane * zb[r]) == OI_OK;
          }
          if (errors[r].empty() && allconv[r]) {
            cs = oi_effdiff_tensor(eh, Ds.data() + (size_t)9 * r);
            if (cs != OI_OK) errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error();
          }
        } while (0);
        if (eh) oi_effdiff_destroy(eh);
      };
      std::vector<std::thread> threads;
      for (int r = 0; r < N; ++r) threads.emplace_back(work, r);
      for (auto& t : threads) t.join();
      for (auto* c : comms)
        if (c) oi_comm_destroy(c);
      for (int r = 0; r < N; ++r)
        if (!errors[r].empty()) throw std::runtime_error("slab " + std::to_string(r) + ": " + errors[r]);
      if (vb >= 1) {
        std::cout << "  z-slabs on a ring of " << N << (h_transport == "local" ? " in-process ranks (one GPU)" : " GPUs") << ":";
        for (int r = 0; r < N; ++r) std::cout << " [" << zb[r] << "," << zb[r + 1] << ")";
        std::cout << (layout0 ? "; CG vectors: lines with an active cell only\n" : "; CG vectors: dense\n");
        for (int k = 0; k < 3; ++k) {
          std::cout << "\n--- Solving for " << what << " chi_" << axis_names[k] << " ---\n  Solver iterations: " << it3[k]
                    << "\n  Final Relative Residual Norm: " << std::scientific << rel3[k] << std::defaultfloat
                    << "\n  Solver Converged Status: " << (conv3[k] ? "Yes" : "No") << "\n";
          if (!conv3[k]) break;
        }
      }
      if (plot_dirs)
        for (int k = 0; k < 3 && conv3[k]; ++k) {
          bool all = true;
          for (int r = 0; r < N; ++r) all = all && got_chi[3 * r + k];
          if (all) write_chi_plotfile((*plot_dirs)[k], k, ph, nx, ny, nz, chi_full[k], vb);
          else std::cerr << "Warning: could not read chi_" << axis_names[k] << " back from every slab; no plotfile\n";
        }
      for (int t = 0; t < 9; ++t) D[t] = Ds[t];
      return allconv[0] != 0;
    };
    auto homogenize = [&](const uint8_t* ph, int nx, int ny, int nz, int vb, const char* what, double D[9],
                          const std::vector<std::string>* plot_dirs) -> bool {
      oi_tort_opts o = hopts;
      o.verbose = vb;
      oi_effdiff* eh = nullptr;
      const auto t_obj = std::chrono::steady_clock::now();
      oi_status est = oi_effdiff_create(&eh, ph, nx, ny, nz, phase_id, &o);
      if (est != OI_OK) throw std::runtime_error(std::string(oi_status_string(est)) + ": " + oi_last_error());
      if (vb >= 1) {
        int compacted = 0;
        int64_t nel = 0;
        (void)oi_effdiff_layout(eh, &compacted, &nel);
        std::cout << "  [wall] solver object (upload, codes, hierarchy): " << since(t_obj) << " s; CG vectors of " << nel << " doubles ("
                  << (compacted ? "lines with an active cell only" : "dense") << ")\n";
      }
      bool ok = true;
      for (int k = 0; k < 3 && ok; ++k) {
        if (vb >= 1) std::cout << "\n--- Solving for " << what << " chi_" << axis_names[k] << " ---\n";
        int conv = 0, iters = 0;
        double rel = 0.0;
        const auto t_solve = std::chrono::steady_clock::now();
        est = oi_effdiff_solve(eh, k, &conv, &iters, &rel);
        if (est != OI_OK) { oi_effdiff_destroy(eh); throw std::runtime_error(std::string(oi_status_string(est)) + ": " + oi_last_error()); }
        if (vb >= 1)
          std::cout << "  Solver iterations: " << iters << "\n  Final Relative Residual Norm: " << std::scientific << rel
                    << std::defaultfloat << "\n  Solver Converged Status: " << (conv ? "Yes" : "No") << "\n  [wall] solve: " << since(t_solve)
                    << " s\n";
        if (!conv) {
          ok = false;
          if (plot_dirs) std::cerr << "Warning: Skipping plotfile write for chi_k because solver did not converge and had active cells.\n";  // :681-684
        } else if (plot_dirs) {
          std::vector<double> chi((size_t)nx * ny * nz);
          if (oi_effdiff_get_chi(eh, k, chi.data()) == OI_OK) write_chi_plotfile((*plot_dirs)[k], k, ph, nx, ny, nz, chi, vb);
          else std::cerr << "Warning: could not read chi_" << axis_names[k] << " back: " << oi_last_error() << "\n";
        }
      }
      if (ok) {
        const auto t_ten = std::chrono::steady_clock::now();
        est = oi_effdiff_tensor(eh, D);
        if (est != OI_OK) { oi_effdiff_destroy(eh); throw std::runtime_error(std::string(oi_status_string(est)) + ": " + oi_last_error()); }
        if (vb >= 1) std::cout << "  [wall] tensor: " << since(t_ten) << " s\n";
      }
      oi_effdiff_destroy(eh);
      return ok;
    };

    if (rev_study) {
      if (verbose >= 1)
        std::cout << "\n--- Starting REV Study (Homogenization Method) for Phase ID " << phase_id << " ---\n"
                  << "  Number of samples per size: " << rev_num_samples << "\n  Target REV sizes: " << rev_sizes
                  << "\n  REV Solver: " << rev_solver << "\n  REV Plotfiles: " << (rev_write_plotfiles ? "Yes" : "No") << std::endl;
      (void)solver_from_string(rev_solver);
      std::vector<int> sizes;
      {
        std::stringstream ss(rev_sizes);
        int v;
        while (ss >> v) sizes.push_back(v);
      }
      if (sizes.empty()) {
        std::cerr << "Warning: REV sizes string is empty or invalid. Skipping REV study.\n";  // :332
      } else {
        std::ofstream csv(results_dir / rev_results_file);
        csv << "SampleNo,SeedX,SeedY,SeedZ,REV_Size_Target,ActualSizeX,ActualSizeY,ActualSizeZ,D_xx,D_yy,D_zz,D_xy,D_xz,D_yz\n";
        std::mt19937 gen(0 + 12345 + rev_num_samples);  // MyProc() + 12345 + num_samples (:341), one process here
        const int dims[3] = {vol.nx, vol.ny, vol.nz};
        std::vector<uint8_t> sub;
        for (int s_idx = 0; s_idx < rev_num_samples; ++s_idx) {
          for (int target : sizes) {
            int lo[3], len[3];
            for (int d = 0; d < 3; ++d) {  // :346-355: uniform seed so that the box fits, else the domain corner
              const int max_coord = dims[d] - 1 - (target - 1);
              if (0 > max_coord || target > dims[d]) {
                lo[d] = 0;
              } else {
                std::uniform_int_distribution<> distr(0, max_coord);
                lo[d] = distr(gen);
              }
              len[d] = std::max(0, std::min(lo[d] + target, dims[d]) - lo[d]);  // box &= domain (:359)
            }
            const int longside = std::max(len[0], std::max(len[1], len[2]));
            if (len[0] <= 0 || len[1] <= 0 || len[2] <= 0 || longside < 8) {  // :361-368
              if (rev_verbose >= 1)
                std::cerr << "Warning: Skipping REV for sample " << s_idx + 1 << " target size " << target
                          << " due to small/empty box after intersection\n";
              continue;
            }
            if (rev_verbose >= 1)
              std::cout << " REV Sample " << s_idx + 1 << ", Target Size " << target << ", Seed Lo (global): (" << lo[0] << ","
                        << lo[1] << "," << lo[2] << "), Actual REV Box (global): ((" << lo[0] << "," << lo[1] << "," << lo[2]
                        << ") (" << lo[0] + len[0] - 1 << "," << lo[1] + len[1] - 1 << "," << lo[2] + len[2] - 1 << "))" << std::endl;
            sub.resize((size_t)len[0] * len[1] * len[2]);
            for (int k = 0; k < len[2]; ++k)
              for (int j = 0; j < len[1]; ++j)
                std::memcpy(&sub[((size_t)k * len[1] + j) * len[0]],
                            &vol.phase[((size_t)(lo[2] + k) * vol.ny + (lo[1] + j)) * vol.nx + lo[0]], (size_t)len[0]);
            double D[9];
            for (double& v : D) v = std::numeric_limits<double>::quiet_NaN();  // :465-466
            bool ok = false;
            try {
              // Diffusion.cpp:432-440: <results>/REV_Sample<s>_Size<len_x>_Dir<k>/effdiff_chi_dir<k> when rev.write_plotfiles is set
              std::vector<std::string> rev_dirs;
              for (int k = 0; k < 3; ++k)
                rev_dirs.push_back((results_dir / ("REV_Sample" + std::to_string(s_idx + 1) + "_Size" + std::to_string(len[0]) + "_Dir" + std::to_string(k))).string());
              ok = homogenize(sub.data(), len[0], len[1], len[2], rev_verbose >= 2 ? rev_verbose : 0, "REV", D, rev_write_plotfiles != 0 ? &rev_dirs : nullptr);
              if (!ok && rev_verbose >= 1) std::cout << "    REV Chi solve FAILED" << std::endl;
            } catch (const std::exception& e) {
              std::cout << "    REV Chi solve EXCEPTION: " << e.what() << std::endl;  // :459-461
            }
            if (!ok) for (double& v : D) v = std::numeric_limits<double>::quiet_NaN();
            csv << s_idx + 1 << "," << lo[0] << "," << lo[1] << "," << lo[2] << "," << target << "," << len[0] << "," << len[1]
                << "," << len[2] << "," << std::fixed << std::setprecision(8) << D[0] << "," << D[4] << "," << D[8] << ","
                << D[1] << "," << D[2] << "," << D[5] << "\n";
            csv.flush();
          }
        }
      }
    }
    if (rev_study && method == "skip_if_rev") {  // :505
      const double secs_r = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      std::cout << std::endl << "Total run time (seconds) = " << secs_r << std::endl;
      return 0;
    }

    if (method == "homogenization") {
      // ---- Diffusion.cpp:509-589: three periodic cell problems chi_k, then the D_eff tensor ----
      if (verbose >= 1) {
        std::cout << "\n--- Full Domain Calculation (" << method << ") using phase " << phase_id << " ---\n";
        std::cout << "\n--- Effective Diffusivity via Homogenization (Full Domain) ---\n";
      }
      (void)solver_from_string(solver_str);  // invalid names abort in the reference (:45-58)
      double D[9];
      if (h_ngpus < 1) throw std::runtime_error("gpu.ngpus must be between 1 and min(64, nz)");
      // Diffusion.cpp:531-545: <results>/FullDomain_chi_<X|Y|Z>/effdiff_chi_dir<k> when write_plotfile is set
      std::vector<std::string> full_dirs;
      for (int k = 0; k < 3; ++k) full_dirs.push_back((results_dir / (std::string("FullDomain_chi_") + axis_names[k])).string());
      const std::vector<std::string>* pd = write_plotfile != 0 ? &full_dirs : nullptr;
      if (h_ngpus > 1 ? homogenize_ring(vol.phase.data(), vol.nx, vol.ny, vol.nz, verbose, "Full Domain", D, pd)
                      : homogenize(vol.phase.data(), vol.nx, vol.ny, vol.nz, verbose, "Full Domain", D, pd)) {
        std::cout << "Full Domain Effective Diffusivity Tensor D_eff / D_material:\n";  // :573-582
        for (int r = 0; r < 3; ++r) {
          std::cout << "  [";
          for (int c = 0; c < 3; ++c) std::cout << std::scientific << std::setprecision(8) << D[3 * r + c] << (c == 2 ? "" : ", ");
          std::cout << "]\n";
        }
        // the reference only prints the tensor; this build also leaves it in the results directory
        std::ofstream tf(results_dir / "Deff_tensor.txt");
        tf << "# D_eff / D_material (homogenization), phase " << phase_id << ", input " << filename << "\n";
        for (int r = 0; r < 3; ++r)
          tf << std::scientific << std::setprecision(9) << D[3 * r] << " " << D[3 * r + 1] << " " << D[3 * r + 2] << "\n";
      } else {
        std::cout << "Full domain D_eff calculation skipped due to chi_k non-convergence.\n";  // :587-589
      }
      write_cathode_files((double)std::count(vol.phase.begin(), vol.phase.end(), (std::uint8_t)phase_id) / (double)std::max<std::size_t>(1, vol.phase.size()));
      const double secs_h = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      std::cout << std::endl << "Total run time (seconds) = " << std::defaultfloat << secs_h << std::endl;
      return 0;
    }
    if (method != "flow_through") {  // any other string falls through every branch of the reference
      std::cerr << "calculation_method = " << method << ": expected 'homogenization' or 'flow_through'.\n";
      return 2;
    }

    // ---- flow_through branch, Diffusion.cpp:591-733 ----
    if (verbose >= 1) std::cout << "\n--- Full Domain Calculation: Tortuosity via Flow-Through ---\n";
    double vlo = -1.0, vhi = 1.0;  // :605-606
    in.query("tortuosity.vlo", vlo);
    in.query("tortuosity.vhi", vhi);

    oi_tort_opts opts;
    oi_tort_default_opts(&opts);
    if (!in.query("hypre.eps", opts.eps)) in.query("hypre_eps", opts.eps);
    // hypre.maxiter counts multigrid-preconditioned FlexGMRES iterations in the reference (default
    // 200); CG needs O(10 N).  It is read for compatibility and honoured only on request.
    int hypre_maxiter = 0, honour = 0;
    if (!in.query("hypre.maxiter", hypre_maxiter)) in.query("hypre_maxiter", hypre_maxiter);
    in.query("gpu.honor_hypre_maxiter", honour);
    if (honour && hypre_maxiter > 0) opts.maxiter = hypre_maxiter;
    in.query("gpu.maxiter", opts.maxiter);
    int tverbose = verbose;
    in.query("tortuosity.verbose", tverbose);
    in.query("tortuosity.remspot_passes", opts.remspot_passes);
    opts.remspot_box_size = box_size;  // the filter runs per box of ba.maxSize(box_size), as in the reference
    // default: the multigrid cycle (the role of SMG in the reference's FlexGMRES, tens of iterations);
    // jacobi / none select the plain single-kernel CG iteration (O(10 N) iterations)
    std::string precond = "mg", kernel = "auto";
    in.query("gpu.precond", precond);
    in.query("gpu.kernel", kernel);
    const std::string pl = lower(precond);
    if (pl == "mg" || pl == "multigrid") opts.precond = OI_PRECOND_MG;
    else if (pl == "jacobi") opts.precond = OI_PRECOND_JACOBI;
    else if (pl == "none") opts.precond = OI_PRECOND_NONE;
    else throw std::runtime_error("gpu.precond must be mg, jacobi or none (got '" + precond + "')");
    const std::string kl = lower(kernel);
    opts.kernel = kl == "unfused" ? OI_KERNEL_UNFUSED
                  : (kl == "fused" ? OI_KERNEL_FUSED
                     : (kl == "fused1" ? OI_KERNEL_FUSED1 : (kl == "lean" ? OI_KERNEL_LEAN : OI_KERNEL_AUTO)));
    if (opts.kernel == OI_KERNEL_LEAN && opts.precond == OI_PRECOND_MG) opts.precond = OI_PRECOND_JACOBI;  // capacity mode: CG
    if (opts.precond == OI_PRECOND_MG && opts.kernel == OI_KERNEL_FUSED1) opts.kernel = OI_KERNEL_AUTO;  // the cycle sits between the two kernels
    in.query("gpu.device", opts.device);
    // gpu.p_storage = f64 (default, every vector fp64 as in the reference) | f32: the search direction p kept as fp32 in HBM -- x, r, q
    // and every operation stay fp64, r stays the exact recurrence residual of x; same iteration counts at 1024^3, the solve 7-9 %
    // shorter (profiles/r04/mg_pcg_p_stored_fp32.txt)
    {
      std::string ps = "f64";
      in.query("gpu.p_storage", ps);
      const std::string psl = lower(ps);
      if (psl == "f32" || psl == "fp32" || psl == "float") opts.p_storage = OI_STORE_F32;
      else if (psl == "f64" || psl == "fp64" || psl == "double") opts.p_storage = OI_STORE_F64;
      else throw std::runtime_error("gpu.p_storage must be f64 or f32 (got '" + ps + "')");
    }
    // gpu.ngpus > 1: one host thread per GPU, z-slabs, RCCL over xGMI (the reference's mpirun -np N);
    // gpu.transport = local keeps all slabs on ONE GPU (in-process transport: a rehearsal of the slab path)
    int ngpus = 1;
    std::string transport = "rccl";
    in.query("gpu.ngpus", ngpus);
    in.query("gpu.transport", transport);
    if (ngpus < 1 || ngpus > 64 || ngpus > vol.nz) throw std::runtime_error("gpu.ngpus must be between 1 and min(64, nz)");
    if (transport != "rccl" && transport != "local") throw std::runtime_error("gpu.transport must be rccl or local");
    if (ngpus > 1 && transport == "rccl" && oi_device_count() < ngpus)
      throw std::runtime_error("gpu.ngpus = " + std::to_string(ngpus) + " but only " + std::to_string(oi_device_count()) +
                               " GPU(s) are visible (gpu.transport = local runs the slabs on one GPU)");

    if (verbose > 0) std::cout << "Calculating Volume Fraction for Phase ID: " << phase_id << "\n";
    long long phase_voxels = 0, total_voxels = 0;
    OpenImpala::volumeFraction(vol.phase.data(), vol.phase.size(), phase_id, phase_voxels, total_voxels);
    const double volume_fraction = total_voxels > 0 ? static_cast<double>(phase_voxels) / static_cast<double>(total_voxels) : 0.0;
    std::cout << "  Volume Fraction = " << std::fixed << std::setprecision(8) << volume_fraction << "\n";
    write_cathode_files(volume_fraction);

    std::string direction_str;
    in.get("direction", direction_str);  // mandatory in this branch (:633)
    const std::string up = upper(direction_str);
    std::vector<OpenImpala::Direction> dirs;
    if (up.find("ALL") != std::string::npos) {
      dirs = {OpenImpala::Direction::X, OpenImpala::Direction::Y, OpenImpala::Direction::Z};
    } else {
      std::stringstream ss(up);
      std::string d;
      while (ss >> d) {
        if (d == "X") dirs.push_back(OpenImpala::Direction::X);
        else if (d == "Y") dirs.push_back(OpenImpala::Direction::Y);
        else if (d == "Z") dirs.push_back(OpenImpala::Direction::Z);
      }
    }
    if (dirs.empty()) std::cerr << "Warning: No valid directions specified in 'direction' input. Skipping tortuosity calculation.\n";

    std::map<std::string, double> results;  // std::map: alphabetical X, Y, Z as in the reference
    const auto solver_type = solver_from_string(solver_str);
    // one upload of the phase bytes for all directions (each direction is still a new solver object:
    // mask, hierarchy, solve -- as in the reference, Diffusion.cpp:655-706)
    struct DevBuf {
      void* p = nullptr;
      ~DevBuf() { if (p) oi_dev_free(p); }
    } dev_phase;
    const bool share_upload = opts.device <= 0 && ngpus == 1;  // oi_dev_malloc allocates on the current (= default) device
    if (!dirs.empty() && share_upload) {
      const auto t_up = std::chrono::steady_clock::now();
      oi_status ds = oi_dev_malloc(&dev_phase.p, vol.phase.size());
      if (ds == OI_OK) ds = oi_dev_upload(dev_phase.p, vol.phase.data(), vol.phase.size());
      if (ds != OI_OK) throw std::runtime_error(std::string(oi_status_string(ds)) + ": " + oi_last_error());
      if (verbose >= 1) std::cout << "  Upload of the phase bytes (once for all directions): " << std::defaultfloat << std::setprecision(4) << since(t_up) << " s\n";
    }
    for (const auto dir : dirs) {
      const std::string dc = dir == OpenImpala::Direction::X ? "X" : (dir == OpenImpala::Direction::Y ? "Y" : "Z");
      if (verbose >= 1) std::cout << "\n--- Solving for Tortuosity in Direction: " << dc << " ---\n";
      double tau = std::numeric_limits<double>::quiet_NaN();
      oi_tort_stats st{};
      std::vector<double> phi;
      std::vector<std::uint8_t> mask;
      bool have_phi = false, have_mask = false;
      const auto t_dir = std::chrono::steady_clock::now();
      double t_create_s = 0.0, t_value_s = 0.0;
      if (ngpus == 1) {
        using TM = OpenImpala::TortuosityMI355;
        std::unique_ptr<TM> solver_ptr(
            share_upload ? new TM(TM::DevicePhase{static_cast<const std::uint8_t*>(dev_phase.p)}, vol.nx, vol.ny, vol.nz,
                                  volume_fraction, phase_id, dir, solver_type, results_path, vlo, vhi, tverbose,
                                  write_plotfile != 0, &opts)
                         : new TM(vol.phase.data(), vol.nx, vol.ny, vol.nz, volume_fraction, phase_id, dir, solver_type,
                                  results_path, vlo, vhi, tverbose, write_plotfile != 0, &opts));
        t_create_s = since(t_dir);
        const auto t_val = std::chrono::steady_clock::now();
        tau = solver_ptr->value();
        t_value_s = since(t_val);
        st = solver_ptr->stats();
        if (write_plotfile != 0 && st.converged) {
          phi.resize(vol.phase.size());
          mask.resize(vol.phase.size());
          have_phi = oi_tort_get_solution(solver_ptr->handle(), phi.data()) == OI_OK;
          have_mask = oi_tort_get_mask(solver_ptr->handle(), mask.data()) == OI_OK;
        }
        if (write_active_mask != 0) {
          std::vector<std::uint8_t> am(vol.phase.size());
          if (oi_tort_get_mask(solver_ptr->handle(), am.data()) == OI_OK) {
            const auto mbase = results_dir / ("active_mask_" + dc);
            std::ofstream(mbase.string() + ".raw", std::ios::binary).write(reinterpret_cast<const char*>(am.data()), static_cast<std::streamsize>(am.size()));
            std::ofstream(mbase.string() + ".txt") << "active_mask uint8 nx=" << vol.nx << " ny=" << vol.ny << " nz=" << vol.nz << " x-fastest (1 = phase " << phase_id
                                                  << " voxel connected to both faces of direction " << dc << ")\n";
            if (verbose >= 1) std::cout << "  Active mask written to " << mbase.string() << ".raw\n";
          } else {
            std::cerr << "Warning: could not read the active mask back: " << oi_last_error() << "\n";
          }
        }
      } else {
        // ---- z-slabs: slab boundaries on multiples of 16 planes where the volume allows it (keeps the
        // multigrid hierarchy slab-partitioned for more levels), one host thread per slab ----
        const int N = ngpus;
        std::vector<int> zb(N + 1, 0);
        {
          const int unit = (vol.nz / N >= 32) ? 16 : 1, units = vol.nz / unit;
          for (int r = 0; r <= N; ++r) zb[r] = (int)((long long)units * r / N) * unit;
          zb[N] = vol.nz;
        }
        std::vector<oi_comm*> comms(N, nullptr);
        unsigned char uid[OI_COMM_ID_BYTES];
        if (transport == "local") {
          const oi_status cs = oi_comm_create_local(comms.data(), N, opts.device);
          if (cs != OI_OK) throw std::runtime_error(std::string(oi_status_string(cs)) + ": " + oi_last_error());
        } else {
          const oi_status cs = oi_comm_unique_id(uid);
          if (cs != OI_OK) throw std::runtime_error(std::string(oi_status_string(cs)) + ": " + oi_last_error());
        }
        if (write_plotfile != 0) { phi.assign(vol.phase.size(), 0.0); mask.assign(vol.phase.size(), 0); }
        std::vector<double> taus(N, std::numeric_limits<double>::quiet_NaN());
        std::vector<oi_tort_stats> stats(N);
        std::vector<std::string> errors(N);
        std::vector<int> got_phi(N, 0);
        const size_t plane = (size_t)vol.nx * vol.ny;
        auto work = [&](int r) {
          oi_tort* h = nullptr;
          do {
            if (transport == "rccl") {
              const oi_status cs = oi_comm_create(&comms[r], uid, N, r, r);
              if (cs != OI_OK) { errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error(); break; }
            }
            oi_tort_opts o = opts;
            o.verbose = r == 0 ? tverbose : 0;
            oi_status cs = oi_tort_create_slab(&h, comms[r], vol.phase.data() + plane * zb[r], 0, vol.nx, vol.ny, vol.nz,
                                               zb[r], zb[r + 1] - zb[r], phase_id, static_cast<int>(dir), vlo, vhi, &o);
            if (cs != OI_OK) { errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error(); break; }
            cs = oi_tort_value(h, 0, &taus[r]);
            if (cs == OI_BAD_ARG || cs == OI_HIP_ERROR || cs == OI_RCCL_ERROR) {
              errors[r] = std::string(oi_status_string(cs)) + ": " + oi_last_error();
              break;
            }
            oi_tort_get_stats(h, &stats[r]);
            if (write_plotfile != 0 && stats[r].converged)
              got_phi[r] = oi_tort_get_solution(h, phi.data() + plane * zb[r]) == OI_OK &&
                           oi_tort_get_mask(h, mask.data() + plane * zb[r]) == OI_OK;
          } while (0);
          if (h) oi_tort_destroy(h);
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < N; ++r) threads.emplace_back(work, r);
        for (auto& t : threads) t.join();
        for (auto* c : comms)
          if (c) oi_comm_destroy(c);
        for (int r = 0; r < N; ++r)
          if (!errors[r].empty()) throw std::runtime_error("slab " + std::to_string(r) + ": " + errors[r]);
        tau = taus[0];
        st = stats[0];
        have_phi = write_plotfile != 0 && std::all_of(got_phi.begin(), got_phi.end(), [](int v) { return v != 0; });
        have_mask = have_phi;
        if (verbose >= 1) {
          std::cout << "  z-slabs over " << N << (transport == "local" ? " in-process ranks (one GPU)" : " GPUs") << ":";
          for (int r = 0; r < N; ++r) std::cout << " [" << zb[r] << "," << zb[r + 1] << ")";
          std::cout << "\n";
        }
      }
      results["Tortuosity_" + dc] = tau;
      if (verbose >= 1) {
        const oi_tort_stats& s_ = st;
        std::cout << "  Active Volume Fraction (percolating phase " << phase_id << "): " << std::defaultfloat
                  << std::setprecision(6) << s_.active_vf << "\n";
        std::cout << "  Solver iterations: " << s_.iterations << "\n";
        std::cout << "  Final Relative Residual Norm: " << std::scientific << s_.final_relres << std::defaultfloat << "\n";
        std::cout << "  Solver Converged Status: " << (s_.converged ? "Yes" : "No") << "\n";
        std::cout << "    Flux In  = " << std::fixed << std::setprecision(8) << s_.flux_in << "\n";
        std::cout << "    Flux Out = " << std::fixed << std::setprecision(8) << s_.flux_out << "\n";
        std::cout << "  Timings [s]: mask " << std::defaultfloat << s_.t_mask_s << ", solve " << s_.t_solve_s << ", flux "
                  << s_.t_flux_s << "\n";
        if (ngpus == 1)
          std::cout << "  Wall clock [s]: solver object (mask, codes, vectors, hierarchy) " << t_create_s << ", value() " << t_value_s
                    << ", this direction incl. teardown " << since(t_dir) << "\n";
      }
      if (have_phi) {
        // the reference's plotfile (TortuosityHypre.cpp:710-749): solution_potential, phase_id, active_mask on the boxes of
        // ba.maxSize(box_size), in AMReX's single-level plotfile layout (apps/Plotfile.H) ...
        const auto base = results_dir / ("tortuosity_solution_" + std::to_string(static_cast<int>(dir)));
        if (have_mask) {
          std::vector<double> phase_d(vol.phase.begin(), vol.phase.end()), mask_d(mask.begin(), mask.end());
          if (!OpenImpala::writeSingleLevelPlotfile(base.string(), vol.nx, vol.ny, vol.nz, {phi.data(), phase_d.data(), mask_d.data()},
                                                    {"solution_potential", "phase_id", "active_mask"}, box_size))
            std::cerr << "Warning: could not write plotfile " << base.string() << "\n";
          else if (verbose >= 1)
            std::cout << "  Plotfile written to " << base.string() << "\n";
        }
        // ... and the raw potential field beside it: f64, x fastest, + a one-line header
        std::ofstream(base.string() + ".raw", std::ios::binary)
            .write(reinterpret_cast<const char*>(phi.data()), static_cast<std::streamsize>(phi.size() * sizeof(double)));
        std::ofstream(base.string() + ".txt") << "solution_potential float64 little-endian nx=" << vol.nx
                                              << " ny=" << vol.ny << " nz=" << vol.nz << " x-fastest\n";
      }
      std::cout << "  >>> Calculated Tortuosity (" << dc << "): " << std::fixed << std::setprecision(8) << tau << " <<<\n";
    }

    // ---- results file, Diffusion.cpp:709-732 ----
    const std::filesystem::path out_path = results_dir / output_filename;
    std::cout << "\nWriting final results to: " << out_path << "\n";
    std::ofstream out(out_path);
    if (out.is_open()) {
      out << "# Tortuosity Calculation Results (Flow-Through Method)\n";
      out << "# Input File: " << filename << "\n";
      out << "# Analysis Phase ID: " << phase_id << "\n";
      out << "# -----------------------------\n";
      out << "VolumeFraction: " << std::fixed << std::setprecision(9) << volume_fraction << "\n";
      for (const auto& kv : results) out << kv.first << ": " << std::fixed << std::setprecision(9) << kv.second << "\n";
    } else {
      std::cerr << "Warning: Could not open output file for writing: " << out_path.string() << "\n";
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::cout << std::endl << "Total run time (seconds) = " << std::defaultfloat << secs << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "Diffusion: " << e.what() << std::endl;  // the reference would amrex::Abort here
    return 1;
  }
  return 0;
}
