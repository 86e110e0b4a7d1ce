// Diabat enumeration and descriptor construction of an MS-EVB force call - ONE implementation for the host and for the device (f2).
// Included by evb.hip behind the definitions of RegionRec / SpecAtom / UnitDesc / CouplingDesc / RecipAtom / SpecPlan.
//
//   plan_enumerate     evb_conduct_proton_transfer_recursive + find_evb_reactive_neighbors (src/ms_evb.f90:498-620, :702-764): depth-first search in the
//                      reference's order (reactive protons in atom order, acceptor molecules in ascending molecule index, basic atoms in atom order), on
//                      the PRINCIPAL topology, as an explicit stack (no recursion on the device)
//   plan_apply_hop     evb_change_data_structures_proton_transfer (:843-932) + reorder_molecule_data_structures (:941-1006) on the compact topology of
//                      the molecules a hop chain touches
//   plan_build_diabat  the evaluation units (old / new side of every hop), the reciprocal-space atoms and the coupling descriptor of one diabat
//
// Everything works on flat fixed-size containers and plain arrays (PlanEnv points at host vectors or at device buffers); errors are codes (plan_error_text).
// The host runs these functions when option evb_device_plan is off - it is then the checker of the device path, which runs the same source in k_evb_plan.
#pragma once

#define PLAN_HD __host__ __device__ inline

enum PlanErr { PE_OK = 0, PE_DONOR_OUTSIDE, PE_SHELL_FULL, PE_MAX_STATES, PE_HOP_OUTSIDE, PE_ACCEPTOR_LARGE, PE_NO_CONJUGATE, PE_REORDER, PE_NO_HEAVY, PE_NO_COUPLING_INDEX, PE_SPEC_FULL,
               PE_HYD_OUTSIDE, PE_CHAIN_FULL, PE_NOT_IN_TOPOLOGY, PE_STAGING, PE_REGION_FULL, PE_DESC_ROOM, PE_PAIRS_FULL };
inline const char* plan_error_text(int e) {
  switch (e) {
    case PE_DONOR_OUTSIDE: return "EVB: donor outside the gathered neighbourhood";
    case PE_SHELL_FULL: return "EVB: more than 64 molecules in a first solvation shell";
    case PE_MAX_STATES: return "Found more diabat states than evb_max_states";
    case PE_HOP_OUTSIDE: return "EVB: proton-hop molecule outside the gathered neighbourhood";
    case PE_ACCEPTOR_LARGE: return "EVB: acceptor molecule too large";
    case PE_NO_CONJUGATE: return "EVB: acceptor molecule type has no conjugate acid";
    case PE_REORDER: return "error in reorder_molecule_data_structures";
    case PE_NO_HEAVY: return "couldn't find atom type in subroutine get_heavy_atom_transfer_acid";
    case PE_NO_COUPLING_INDEX: return "couldn't find index in subroutine 'get_index_atom_set' (diabat coupling parameters)";
    case PE_SPEC_FULL: return "EVB: more special atoms than MAX_SPEC";
    case PE_HYD_OUTSIDE: return "EVB: hydronium outside region";
    case PE_CHAIN_FULL: return "EVB: more molecules in a hop chain than MAX_CHAIN + 1";
    case PE_NOT_IN_TOPOLOGY: return "EVB: atom not in the hop-chain topology";
    case PE_STAGING: return "EVB descriptor staging too small";
    case PE_REGION_FULL: return "EVB neighbourhood buffer too small";
    case PE_DESC_ROOM: return "EVB: descriptor buffer too small for the device-side plan";
    case PE_PAIRS_FULL: return "EVB: more than 32 reactive pairs around one donor molecule";
    default: return "EVB plan: unknown error";
  }
}

// static tables of the enumeration (molecule types, conjugates, coupling parameters): filled once per context from the host tables, uploaded once
struct PlanTables {
  int nmt, max_chain, rank, n_ranks;
  int mt_natom[MAX_MOLTYPES], mt_atype[MAX_MOLTYPES][MAX_MOL_ATOMS], mt_nbond[MAX_MOLTYPES], mt_bond[MAX_MOLTYPES][MAX_BONDED][2];
  int reactive_p[MAX_MOLTYPES][MAX_MOL_ATOMS], reactive_b[MAX_MOLTYPES][MAX_MOL_ATOMS];
  int conj_mol[MAX_MOLTYPES], proton_type[MAX_MOLTYPES], heavy_type[MAX_MOLTYPES], conj_atype[MAX_TYPES];
  double atype_chg[MAX_TYPES], e_ref[MAX_MOLTYPES], exch_proton[MAX_MOLTYPES][MAX_MOLTYPES];
  int n_cp, cp_types[MAX_INTERACTION][3], cp_func[MAX_INTERACTION];
  double cp_par[MAX_INTERACTION][10];
  double box[3], inv_box[3];      // inv_box = 1 / box, rounded once on the host (the same bits the division in the loop gave)
};
// what a call works on: the principal topology (arrays of the context) and the neighbourhood record of the hydronium, addressed in ascending molecule order
struct PlanEnv {
  const PlanTables* tb;
  const int* mol_type; const int* mol_natom; const int* mol_atoms; const int* atype; const double* chg; const double* mass;
  const RegionRec* rec; const int* order; int n_region;      // rec[order[k]].mol ascending in k
  int hydronium;
  // device only (nullptr on the host): fast copies of the records' molecule ids and centres of mass in molecule order (LDS), and the reactive pairs of ALL record
  // molecules, tabulated by the whole workgroup before one thread walks the tree: pair_cnt[a] pairs {proton, acceptor molecule, basic atom} at pair_tab[3 PLAN_MAX_PAIRS a ..],
  // pair_cnt[a] < 0: - error code of plan_donor_pairs (raised only if the search expands that molecule)
  const int* sorted_mol; const double* sorted_com; const int* pair_tab; const int* pair_cnt;
};

struct PMol { int id, type, natom; int atoms[MAX_MOL_ATOMS]; double com[3]; };
struct PAtom { int type; double q; double x[3]; int mol; };
struct PTopo {      // the molecules touched so far by a hop chain, in the topology reached so far (HTopo of rounds 1-4)
  int n_mols, n_atoms, hydronium;
  PMol mols[MAX_CHAIN + 1];
  int akey[MAX_SPEC]; PAtom aval[MAX_SPEC];
  PLAN_HD PAtom* find(int a) { for (int i = 0; i < n_atoms; ++i) if (akey[i] == a) return &aval[i]; return nullptr; }
  PLAN_HD const PAtom* find(int a) const { for (int i = 0; i < n_atoms; ++i) if (akey[i] == a) return &aval[i]; return nullptr; }
};
struct PWhere { int n; int key[MAX_SPEC], val[MAX_SPEC]; PLAN_HD int at(int a) const { for (int i = 0; i < n; ++i) if (key[i] == a) return val[i]; return -1; } };
struct PlanWork { PTopo P, T; PWhere where; };      // work space of one diabat's construction (device: global memory, one per thread - not 5 KB of private memory per lane)

PLAN_HD void plan_pbc_shift(const PlanTables* tb, const double* ri, const double* rj, double* sh) { for (int d = 0; d < 3; ++d) sh[d] = floor((rj[d] - ri[d]) * tb->inv_box[d] + 0.5) * tb->box[d]; }
PLAN_HD int plan_find_region_index(const PlanEnv& E, int mol) {      // position in the molecule order, -1: not in the record
  int lo = 0, hi = E.n_region;
  if (E.sorted_mol) {
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (E.sorted_mol[mid] < mol) lo = mid + 1; else hi = mid; }
    return (lo < E.n_region && E.sorted_mol[lo] == mol) ? lo : -1;
  }
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (E.rec[E.order[mid]].mol < mol) lo = mid + 1; else hi = mid; }
  return (lo < E.n_region && E.rec[E.order[lo]].mol == mol) ? lo : -1;
}
PLAN_HD const RegionRec* plan_find_region(const PlanEnv& E, int mol) { const int k = plan_find_region_index(E, mol); return k < 0 ? nullptr : &E.rec[E.order[k]]; }
// is molecule j (record rj) in the first solvation shell of molecule i (record ri)?  evb_first_solvation_cutoff = 5 (src/glob_v.f90:54); sh = the periodic shift used
PLAN_HD bool plan_in_shell(const PlanTables* tb, const double* ci, const double* cj, double* sh) {
  double dc[3]; plan_pbc_shift(tb, ci, cj, sh);
  for (int d = 0; d < 3; ++d) dc[d] = cj[d] - ci[d] - sh[d];
  return !(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2] >= 25.0);
}
PLAN_HD int plan_ensure_mol(const PlanEnv& E, PTopo& t, int m, int* err) {      // index in t.mols, -1: molecule outside the record (err untouched), -2: container full
  for (int k = 0; k < t.n_mols; ++k) if (t.mols[k].id == m) return k;
  const RegionRec* r = plan_find_region(E, m);
  if (!r) return -1;
  if (t.n_mols >= MAX_CHAIN + 1) { *err = PE_CHAIN_FULL; return -2; }
  PMol& hm = t.mols[t.n_mols];
  hm.id = m; hm.type = E.mol_type[m]; hm.natom = E.mol_natom[m];
  for (int k = 0; k < hm.natom; ++k) {
    const int a = E.mol_atoms[(size_t)m * MAX_MOL_ATOMS + k];
    hm.atoms[k] = a;
    PAtom* pa = t.find(a);
    if (!pa) { if (t.n_atoms >= MAX_SPEC) { *err = PE_SPEC_FULL; return -2; } t.akey[t.n_atoms] = a; pa = &t.aval[t.n_atoms++]; }
    pa->type = E.atype[a]; pa->q = E.chg[a]; pa->mol = m;
    for (int d = 0; d < 3; ++d) pa->x[d] = r->x[k][d];
  }
  for (int d = 0; d < 3; ++d) hm.com[d] = r->com[d];
  return t.n_mols++;
}
PLAN_HD void plan_mol_com(const PlanEnv& E, const PTopo& t, PMol& m) {
  double s[3] = {0, 0, 0}, mt = 0;
  for (int k = 0; k < m.natom; ++k) { const PAtom* a = t.find(m.atoms[k]); const double ma = E.mass[m.atoms[k]]; for (int d = 0; d < 3; ++d) s[d] = s[d] + a->x[d] * ma; mt = mt + ma; }
  for (int d = 0; d < 3; ++d) m.com[d] = s[d] / mt;
}
PLAN_HD int plan_heavy_local_acid(const PlanTables* tb, int acid_type) {   // get_heavy_atom_transfer_acid (src/ms_evb.f90:2917)
  for (int i = 0; i < tb->mt_natom[acid_type]; ++i) if (tb->mt_atype[acid_type][i] == tb->heavy_type[acid_type]) return i;
  return -1;
}
// evb_change_data_structures_proton_transfer (src/ms_evb.f90:843-932) on the compact topology
PLAN_HD int plan_apply_hop(const PlanEnv& E, PTopo& t, int md, int ip_local, int ma, int iheavy_local, int* mw_atoms = nullptr, int* mw_n = nullptr) {
  const PlanTables* tb = E.tb;
  int err = PE_OK;
  const int kd = plan_ensure_mol(E, t, md, &err), ka0 = plan_ensure_mol(E, t, ma, &err);
  if (kd < 0 || ka0 < 0) return err != PE_OK ? err : PE_HOP_OUTSIDE;
  PMol& D = t.mols[kd];
  PMol& A = t.mols[ka0];
  const int gp = D.atoms[ip_local];
  for (int k = ip_local; k < D.natom - 1; ++k) D.atoms[k] = D.atoms[k + 1];
  D.natom -= 1;
  if (A.natom >= MAX_MOL_ATOMS) return PE_ACCEPTOR_LARGE;
  A.atoms[A.natom] = gp; A.natom += 1;
  t.find(gp)->mol = ma;
  t.hydronium = ma;
  if (mw_atoms) { *mw_n = A.natom; for (int i = 0; i < A.natom; ++i) mw_atoms[i] = A.atoms[i]; }      // (what k_evb_refresh replays on the next steps' positions)
  for (int i = 1; i < A.natom; ++i) {    // make_molecule_whole (src/general_routines.f90:1065)
    PAtom& prev = *t.find(A.atoms[i - 1]);
    PAtom& cur = *t.find(A.atoms[i]);
    double sh[3]; plan_pbc_shift(tb, prev.x, cur.x, sh);
    if (fabs(sh[0]) > 1e-6 || fabs(sh[1]) > 1e-6 || fabs(sh[2]) > 1e-6) for (int d = 0; d < 3; ++d) cur.x[d] = prev.x[d] + (cur.x[d] - prev.x[d] - sh[d]);
  }
  plan_mol_com(E, t, D); plan_mol_com(E, t, A);
  const int base_type = A.type, acid_type = tb->conj_mol[base_type];
  if (acid_type < 0) return PE_NO_CONJUGATE;
  t.find(gp)->type = tb->proton_type[acid_type];
  for (int i = 0; i < A.natom; ++i) {
    PAtom& a = *t.find(A.atoms[i]);
    const int tn = (i != A.natom - 1) ? tb->conj_atype[a.type] : a.type;
    a.type = tn; a.q = tb->atype_chg[tn];
  }
  t.find(A.atoms[iheavy_local])->type = tb->heavy_type[acid_type];
  for (int i = 0; i < D.natom; ++i) { PAtom& a = *t.find(D.atoms[i]); const int tn = tb->conj_atype[a.type]; a.type = tn; a.q = tb->atype_chg[tn]; }
  D.type = tb->conj_mol[D.type];
  A.type = acid_type;
  for (int i = 0; i < tb->mt_natom[acid_type]; ++i) {      // reorder_molecule_data_structures (src/ms_evb.f90:941)
    if (tb->mt_atype[acid_type][i] != t.find(A.atoms[i])->type) {
      int idx = -1;
      for (int j = i + 1; j < A.natom; ++j) if (tb->mt_atype[acid_type][i] == t.find(A.atoms[j])->type) { idx = j; break; }
      if (idx < 0) return PE_REORDER;
      const int keep = A.atoms[idx];
      for (int j = idx; j > i; --j) A.atoms[j] = A.atoms[j - 1];
      A.atoms[i] = keep;
    }
  }
  return PE_OK;
}
PLAN_HD void plan_fill_spec(const PTopo& t, SpecAtom* spec, int* mols, int& n_spec, int& n_mols, PWhere& where) {
  n_spec = 0; n_mols = 0; where.n = 0;
  for (int im = 0; im < t.n_mols; ++im) {
    const PMol& m = t.mols[im];
    mols[n_mols++] = m.id;
    for (int k = 0; k < m.natom; ++k) {
      const PAtom* a = t.find(m.atoms[k]);
      SpecAtom& s = spec[n_spec];
      s.idx = m.atoms[k]; s.type = a->type; s.mol = m.id; s.q = a->q; s.pad = 0;
      for (int d = 0; d < 3; ++d) s.x[d] = a->x[d];
      where.key[where.n] = m.atoms[k]; where.val[where.n++] = n_spec++;
    }
  }
}
PLAN_HD void plan_zero(void* p, size_t n) {      // (descriptors are multiples of 8 bytes, 8-byte aligned)
  if ((((size_t)p | n) & 7) == 0) { unsigned long long* w = (unsigned long long*)p; for (size_t k = 0; k < n / 8; ++k) w[k] = 0ull; }
  else { unsigned char* b = (unsigned char*)p; for (size_t k = 0; k < n; ++k) b[k] = 0; }
}
PLAN_HD int plan_make_unit(const PlanEnv& E, const PTopo& t, int donor, int acceptor, double sign, int slot, int do_rs, int do_rep, UnitDesc& u, PWhere& where, bool zeroed = false) {
  if (!zeroed) plan_zero(&u, sizeof(u));      // (device: the whole descriptor block was zero-filled by the workgroup - 270 eight-byte stores from ONE lane cost 27 us per descriptor)
  u.sign = sign; u.slot = slot; u.do_rs = do_rs; u.do_rep = do_rep; u.donor = donor; u.acceptor = acceptor; u.hyd = t.hydronium;
  plan_fill_spec(t, u.spec, u.mols, u.n_spec, u.n_mols, where);
  for (int im = 0; im < t.n_mols; ++im) {
    const PMol& m = t.mols[im];
    if (do_rs && m.id == donor) { u.nD = m.natom; u.mt_D = m.type; for (int k = 0; k < m.natom; ++k) u.D_at[k] = where.at(m.atoms[k]); }
    if (do_rs && m.id == acceptor) { u.nA = m.natom; u.mt_A = m.type; for (int k = 0; k < m.natom; ++k) u.A_at[k] = where.at(m.atoms[k]); }
    if (m.id == t.hydronium) {
      u.nH = m.natom;
      for (int k = 0; k < m.natom; ++k) u.H_at[k] = where.at(m.atoms[k]);
      const int hl = plan_heavy_local_acid(E.tb, m.type);
      if (hl < 0) return PE_NO_HEAVY;
      u.heavy = u.H_at[hl];
      u.type_H = t.find(m.atoms[m.natom - 1])->type;
    }
  }
  return PE_OK;
}

// slots of the force stores (diabat d >= 1 lives on rank (d - 1) mod R): diagonal store 1 + R S slots (0 = principal), off-diagonal store R S
PLAN_HD int plan_slots_per_rank(const PlanTables* tb, int D) { return D <= 1 ? 0 : (D - 1 + tb->n_ranks - 1) / tb->n_ranks; }
PLAN_HD int plan_off_slot(const PlanTables* tb, int D, int d) { return ((d - 1) % tb->n_ranks) * plan_slots_per_rank(tb, D) + (d - 1) / tb->n_ranks; }
PLAN_HD int plan_diag_slot(const PlanTables* tb, int D, int d) { return d == 0 ? 0 : 1 + plan_off_slot(tb, D, d); }
PLAN_HD bool plan_owned(const PlanTables* tb, int d) { return d == 0 || ((d - 1) % tb->n_ranks) == tb->rank; }

// ---- enumeration ----------------------------------------------------------------------------------------------------------------------------
constexpr int PLAN_MAX_PAIRS = 32;
// find_evb_reactive_neighbors (src/ms_evb.f90:702-764) for EVERY reactive proton of the record molecule at position a of the molecule order: the pairs
// {proton ia, acceptor molecule, basic atom ja} within evb_reactive_pair_distance, acceptors from the first solvation shell (centres of mass within
// evb_first_solvation_cutoff), in the reference's loop order (protons in atom order, molecules ascending, basic atoms in atom order).
// the first solvation shell of the record molecule at position a of the molecule order: positions (ascending) of the molecules whose centre of mass lies within
// evb_first_solvation_cutoff = 5 A (src/glob_v.f90:54); returns their number (more than 64: the reference's own limit, only the first 64 are stored)
PLAN_HD int plan_shell(const PlanEnv& E, int a, int* near) {
  const double* ci = E.sorted_com ? E.sorted_com + 3 * a : E.rec[E.order[a]].com;
  int n = 0;
  for (int k = 0; k < E.n_region; ++k) {
    if (k == a) continue;
    const double* cj = E.sorted_com ? E.sorted_com + 3 * k : E.rec[E.order[k]].com;
    double sh[3];
    if (!plan_in_shell(E.tb, ci, cj, sh)) continue;
    if (n < 64) near[n] = k;
    ++n;
  }
  return n;
}
PLAN_HD int plan_donor_pairs(const PlanEnv& E, int a, const int* near, int n_near, int (*pairs)[3], int* n_out) {
  const PlanTables* tb = E.tb;
  if (n_near > 64) return PE_SHELL_FULL;
  const RegionRec& ri = E.rec[E.order[a]];
  const int mt = E.mol_type[ri.mol], na = E.mol_natom[ri.mol];
  int n = 0;
  for (int ia = 0; ia < na; ++ia) {
    if (tb->reactive_p[mt][ia] != 1) continue;
    for (int q = 0; q < n_near; ++q) {      // ascending molecule index == the reference's j_mole loop
      const RegionRec& rj = E.rec[E.order[near[q]]];
      double sh[3]; plan_pbc_shift(tb, ri.com, rj.com, sh);
      const int mj = E.mol_type[rj.mol];
      for (int ja = 0; ja < rj.natom; ++ja) {
        if (tb->reactive_b[mj][ja] != 1) continue;
        double r[3]; for (int d = 0; d < 3; ++d) r[d] = rj.x[ja][d] - ri.x[ia][d] - sh[d];
        if (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] >= 6.25) continue;           // evb_reactive_pair_distance = 2.5 (src/glob_v.f90:55)
        if (n >= PLAN_MAX_PAIRS) return PE_PAIRS_FULL;
        pairs[n][0] = ia; pairs[n][1] = rj.mol; pairs[n][2] = ja; ++n;
      }
    }
  }
  *n_out = n;
  return PE_OK;
}
struct PlanFrame {      // one level of evb_conduct_proton_transfer_recursive: the donor, the diabat it was reached in, its reactive pairs, and how far the loop over them has got
  int mol, diabat, count, dq, dnew, n_pairs, ip, ready;
  int pairs[PLAN_MAX_PAIRS][3];
};
// st: MAX_CHAIN + 2 frames of scratch.  plog / parent / D: the tree; plan: donors and pairs of the speculative step's tree check (fits: everything fitted its fixed tables)
PLAN_HD int plan_enumerate(const PlanEnv& E, PlanFrame* st, int (*plog)[MAX_CHAIN][5], int* parent, int* D_out, SpecPlan& plan, int* fits) {
  const PlanTables* tb = E.tb;
  int D = 1, sp = 0;
  plan.n_donors = 0; plan.n_pairs_total = 0; plan.hyd = E.hydronium;
  st[sp].mol = E.hydronium; st[sp].diabat = 0; st[sp].ready = 0; ++sp;
  while (sp > 0) {
    PlanFrame& f = st[sp - 1];
    if (!f.ready) {
      int count = 0;
      for (int h = 0; h < tb->max_chain; ++h) { if (plog[f.diabat][h][0] < 0) break; count++; }
      if (count >= tb->max_chain) { --sp; continue; }
      const int a = plan_find_region_index(E, f.mol);
      if (a < 0) return PE_DONOR_OUTSIDE;
      f.count = count; f.dq = -1; f.dnew = 0;      // this molecule's entry among the expanded donors of the plan (its pairs are recorded once)
      for (int q = 0; q < plan.n_donors; ++q) if (plan.donor_mol[q] == f.mol) f.dq = q;
      if (f.dq < 0) { if (plan.n_donors < SPEC_MAX_DONORS) { f.dq = plan.n_donors++; plan.donor_mol[f.dq] = f.mol; plan.donor_np[f.dq] = 0; f.dnew = 1; } else *fits = 0; }
      if (E.pair_cnt) {      // tabulated (device)
        const int cnt = E.pair_cnt[a];
        if (cnt < 0) return -cnt;
        for (int q = 0; q < cnt; ++q) for (int z = 0; z < 3; ++z) f.pairs[q][z] = E.pair_tab[3 * (PLAN_MAX_PAIRS * a + q) + z];
        f.n_pairs = cnt;
      } else { int near[64]; const int nn = plan_shell(E, a, near); const int pe = plan_donor_pairs(E, a, near, nn, f.pairs, &f.n_pairs); if (pe) return pe; }
      f.ip = 0; f.ready = 1;
    }
    const int mt = E.mol_type[f.mol];
    bool pushed = false;
    while (f.ip < f.n_pairs) {
      const int ia = f.pairs[f.ip][0], mj = f.pairs[f.ip][1], ja = f.pairs[f.ip][2];
      ++f.ip;
      if (f.dnew) { if (plan.donor_np[f.dq] < SPEC_MAX_PAIRS) { int* pp = plan.donor_pair[f.dq][plan.donor_np[f.dq]++]; pp[0] = ia; pp[1] = mj; pp[2] = ja; plan.n_pairs_total++; } else *fits = 0; }
      if (D >= MAX_DIABATS) return PE_MAX_STATES;
      const int d = D++;
      parent[d] = f.diabat;
      for (int h = 0; h < f.count; ++h) for (int q5 = 0; q5 < 5; ++q5) plog[d][h][q5] = plog[f.diabat][h][q5];
      int heavy = -1;
      for (int b = 0; b < tb->mt_nbond[mt]; ++b) { if (tb->mt_bond[mt][b][0] == ia) heavy = tb->mt_bond[mt][b][1]; else if (tb->mt_bond[mt][b][1] == ia) heavy = tb->mt_bond[mt][b][0]; }
      plog[d][f.count][0] = f.mol; plog[d][f.count][1] = ia; plog[d][f.count][2] = heavy; plog[d][f.count][3] = mj; plog[d][f.count][4] = ja;
      if (mj != E.hydronium) {      // descend; this level's loop goes on behind the pair just taken (a transfer onto the current hydronium ends the branch, flag_cycle :573)
        st[sp].mol = mj; st[sp].diabat = d; st[sp].ready = 0; ++sp;
        pushed = true;
        break;
      }
    }
    if (!pushed) --sp;
  }
  *D_out = D;
  return PE_OK;
}

// atoms of the molecules diabat d's hop chain touches (hops move atoms between them, the total stays): its number of reciprocal-space atoms
PLAN_HD int plan_chain_atoms(const PlanEnv& E, int d, const int (*plog)[MAX_CHAIN][5], int* nhop_out) {
  int mols[MAX_CHAIN + 1], nm = 0, n = 0, nhop = 0;
  mols[nm++] = E.hydronium;
  for (int h = 0; h < E.tb->max_chain; ++h) {
    if (plog[d][h][0] < 0) break;
    nhop = h + 1;
    const int ma = plog[d][h][3];
    bool seen = false;
    for (int k = 0; k < nm; ++k) seen = seen || mols[k] == ma;
    if (!seen) mols[nm++] = ma;
  }
  for (int k = 0; k < nm; ++k) n += E.mol_natom[mols[k]];
  *nhop_out = nhop;
  return n;
}

// ---- one diabat's descriptors ---------------------------------------------------------------------------------------------------------------
// W.P: the principal topology with the hydronium in it (plan_ensure_mol); W.T, W.where: scratch.  Writes units[u0 .. u0 + 2 nhop), ras[r0 ..) (returns their number in *n_ras), the coupling cp
// (slot, geometry, parameters), the plan's unit_d / unit_k / mw_atoms / mw_n of this diabat, and the reference-energy difference.
PLAN_HD int plan_build_diabat(const PlanEnv& E, PlanWork& W, int d, int D, const int (*plog)[MAX_CHAIN][5], UnitDesc* units, int u0, RecipAtom* ras, int r0, int grid, CouplingDesc& cp,
                              SpecPlan& plan, double* eref_delta, int* n_ras, int* nhop_out, bool zeroed = false) {
  const PlanTables* tb = E.tb;
  PTopo& T = W.T; PWhere& where = W.where;
  T = W.P;
  const int slot = plan_diag_slot(tb, D, d);
  int md = E.hydronium, ma = E.hydronium, nhop = 0, err = PE_OK;
  for (int h = 0; h < tb->max_chain; ++h) if (plog[d][h][0] >= 0) nhop = h + 1;
  double er = 0.0;
  int iu = u0;
  for (int h = 0; h < tb->max_chain; ++h) {
    if (plog[d][h][0] < 0) break;
    md = T.hydronium; ma = plog[d][h][3];
    if (plan_ensure_mol(E, T, md, &err) < 0 || plan_ensure_mol(E, T, ma, &err) < 0) return err != PE_OK ? err : PE_HOP_OUTSIDE;
    if (iu + 1 >= MAX_UNITS) return PE_STAGING;
    { const int rc = plan_make_unit(E, T, md, ma, -1.0, slot, 1, 0, units[iu], where, zeroed); if (rc) return rc; plan.unit_d[iu] = d; plan.unit_k[iu] = h; ++iu; }
    int typ_old = -1; for (int k = 0; k < T.n_mols; ++k) if (T.mols[k].id == T.hydronium) typ_old = T.mols[k].type;
    { const int rc = plan_apply_hop(E, T, md, plog[d][h][1], ma, plog[d][h][4], plan.mw_atoms[d][h], &plan.mw_n[d][h]); if (rc) return rc; }
    { const int rc = plan_make_unit(E, T, md, ma, +1.0, slot, 1, h == nhop - 1 ? 1 : 0, units[iu], where, zeroed); if (rc) return rc; plan.unit_d[iu] = d; plan.unit_k[iu] = h + 1; ++iu; }
    int typ_new = -1; for (int k = 0; k < T.n_mols; ++k) if (T.mols[k].id == T.hydronium) typ_new = T.mols[k].type;
    er += tb->e_ref[typ_new] - tb->e_ref[typ_old];
  }
  *eref_delta = er; *nhop_out = nhop;
  // reciprocal-space patches: every atom of a touched molecule
  int nr = 0;
  for (int im = 0; im < T.n_mols; ++im) for (int k = 0; k < T.mols[im].natom; ++k) {
    const int a = T.mols[im].atoms[k]; const PAtom* ha = T.find(a);
    RecipAtom r; r.grid = grid; r.slot = slot; r.idx = a; r.pad = 0; r.q_new = ha->q; r.q_old = E.chg[a];
    for (int q = 0; q < 3; ++q) r.x_new[q] = ha->x[q];
    ras[r0 + nr++] = r;
  }
  *n_ras = nr;
  // off-diagonal element with the parent diabat, in this (acceptor) topology (evb_diabatic_coupling, :1021)
  if (!zeroed) plan_zero(&cp, sizeof(cp));
  cp.slot = plan_off_slot(tb, D, d);
  plan_fill_spec(T, cp.spec, cp.mols, cp.n_spec, cp.n_mols, where);
  const PMol *Dm = nullptr, *Am = nullptr;
  for (int k = 0; k < T.n_mols; ++k) { if (T.mols[k].id == md) Dm = &T.mols[k]; if (T.mols[k].id == ma) Am = &T.mols[k]; for (int q = 0; q < 3; ++q) cp.mol_com[k][q] = T.mols[k].com[q]; }
  cp.nD = Dm->natom; cp.nA = Am->natom;
  for (int k = 0; k < Dm->natom; ++k) cp.D_at[k] = where.at(Dm->atoms[k]);
  for (int k = 0; k < Am->natom; ++k) cp.A_at[k] = where.at(Am->atoms[k]);
  double mdt = 0, mat = 0;
  for (int k = 0; k < Dm->natom; ++k) mdt = mdt + E.mass[Dm->atoms[k]];
  for (int k = 0; k < Am->natom; ++k) mat = mat + E.mass[Am->atoms[k]];
  double sha[3]; plan_pbc_shift(tb, Dm->com, Am->com, sha);      // zundel_r_com (src/ms_evb.f90:2946)
  for (int q = 0; q < 3; ++q) { const double rca = Dm->com[q] + (Am->com[q] - Dm->com[q] - sha[q]); cp.zc[q] = (mdt * Dm->com[q] + mat * rca) / (mdt + mat); cp.shiftd[q] = 0.0; cp.shifta[q] = sha[q]; }
  cp.q_transfer = tb->exch_proton[Am->type][Dm->type];
  const int i_don = plan_heavy_local_acid(tb, tb->conj_mol[Dm->type]), i_acc = plan_heavy_local_acid(tb, Am->type);
  if (i_don < 0 || i_acc < 0) return PE_NO_HEAVY;
  cp.gO1 = where.at(Dm->atoms[i_don]); cp.gO2 = where.at(Am->atoms[i_acc]); cp.gH = where.at(Am->atoms[Am->natom - 1]);
  const int t1 = cp.spec[cp.gO1].type, t2 = cp.spec[cp.gO2].type, t3 = cp.spec[cp.gH].type;
  int ic = -1;
  for (int k = 0; k < tb->n_cp; ++k) if (tb->cp_types[k][0] == t1 && tb->cp_types[k][1] == t2 && tb->cp_types[k][2] == t3) { ic = k; break; }
  if (ic < 0) return PE_NO_COUPLING_INDEX;
  cp.ftype = tb->cp_func[ic];
  for (int k = 0; k < 10; ++k) cp.par[k] = tb->cp_par[ic][k];
  return PE_OK;
}
