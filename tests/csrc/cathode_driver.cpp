// test driver: cathode_driver <vf_solid> <radius> <conductivity> <cmax> <dandeliion file> <pybamm file>
#include <cstdlib>

#include "../../apps/CathodeWrite.H"

int main(int argc, char** argv) {
  if (argc != 7) return 2;
  OpenImpala::CathodeParams p;
  p.volume_fraction_solid = std::atof(argv[1]);
  p.particle_radius = std::atof(argv[2]);
  p.active_material_conductivity = std::atof(argv[3]);
  p.max_concentration = std::atof(argv[4]);
  const OpenImpala::CathodeWrite w(p);
  const bool a = w.writeDandeLiionParameters(argv[5]);
  const bool b = w.writePyBammParameters(argv[6]);
  return (a ? 0 : 1) | (b ? 0 : 4);
}
