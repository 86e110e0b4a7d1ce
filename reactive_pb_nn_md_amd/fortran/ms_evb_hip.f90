!==============================================================================
! module ms_evb for a GPU-backed build: everything of the reference's module
! (src/ms_evb.f90 - the EVB initialisation, the .top EVB parsers, the topology
! hand-over, the printing) stays the reference's own code, only the hot entry
! point is replaced:
!
!   ms_evb_calculate_total_force_energy( system_data, molecule_data, atom_data, verlet_list_data, PME_data, &
!                                        file_io_data, n_output, integrator_data, trajectory_step )     (src/ms_evb.f90:181)
!
! with the reference's nine-argument list, so md_integration.f90 and
! initialize_routines.f90 compile unchanged against it.
!
! Build recipe (done by `make link_check`, see INTEGRATION.md): the reference's
! ms_evb.f90 is compiled with ONE token changed on the way, its module name,
!     sed 's/^\(\s*\)\(end \)\?module ms_evb\s*$/\1\2module ms_evb_ref/I'
! and this file provides `module ms_evb` on top of it.  Fortran re-exports what a
! module `use`s, so every other public entity of the reference's module reaches
! the callers through this one; the hot procedure is hidden by a rename.
!==============================================================================
module ms_evb
  use ms_evb_ref, ms_evb_calculate_total_force_energy_cpu => ms_evb_calculate_total_force_energy
  use total_energy_forces, only : msevb_hip_ms_evb_force_energy, msevb_hip_remap
  implicit none
  integer, save, private :: i_step_hip = 0      ! diagonalize_evb_hamiltonian's own step counter (src/ms_evb.f90:265,:347)

contains

  subroutine ms_evb_calculate_total_force_energy( system_data, molecule_data, atom_data, verlet_list_data, PME_data, file_io_data, n_output, integrator_data, trajectory_step  )
    type(system_data_type), intent(inout)      :: system_data
    type(molecule_data_type), dimension(:), intent(inout)    :: molecule_data
    type(atom_data_type), intent(inout)        :: atom_data
    type(verlet_list_data_type), intent(inout) :: verlet_list_data
    type( integrator_data_type ), intent(in)    :: integrator_data
    type(PME_data_type)   , intent(inout)      :: PME_data
    type(file_io_data_type), intent(in)        :: file_io_data
    integer, intent(in)                        :: n_output
    integer, intent(in)                        :: trajectory_step

    integer :: i_mole_hydronium, new_i_mole_hydronium, principle_diabat
    real*8 :: ground_state_eigenvector(evb_max_states)

    if ( n_hydronium_molecules > 1 ) then
       stop "can't have more than 1 hydronium"
    end if
    i_mole_hydronium = hydronium_molecule_index(1)

    ! construct_evb_hamiltonian + diagonalize_evb_hamiltonian (src/ms_evb.f90:375, :242) on the GPU; the module variables the
    ! reference's own routines read afterwards are filled from the library's results
    call msevb_hip_ms_evb_force_energy( system_data, molecule_data, atom_data, verlet_list_data, PME_data, diabat_index, evb_diabat_proton_log, &
         evb_diabat_coupling_matrix, evb_hamiltonian, ground_state_eigenvector, principle_diabat, new_i_mole_hydronium )

    !**** print evb trajectory info if requested, as diagonalize_evb_hamiltonian does (src/ms_evb.f90:329-344)
    Select Case( print_ms_evb_data )
    Case("yes")
       if ( new_i_mole_hydronium /= i_mole_hydronium ) then
          write(file_io_data%ofile_hop_file_h,*) "step ", trajectory_step
          write(file_io_data%ofile_hop_file_h,*) "proton hop from ", i_mole_hydronium, " to ", new_i_mole_hydronium
       end if
       if ( mod( i_step_hip, n_output ) == 0 ) then
          if ( ( restart_trajectory .eq. "no" ) .or. ( i_step_hip > 0 ) ) then
             call print_evb_trajectory_data( ground_state_eigenvector(1:diabat_index) , i_mole_hydronium, file_io_data )
          end if
       endif
    End Select
    i_step_hip = i_step_hip + 1

    ! principal-diabat switch (src/ms_evb.f90:218-227): the host arrays are re-topologised by the reference's OWN routine (it moves the
    ! proton's array slot and every atom in between, forces and velocities included); the device copy, which never moves atoms, stays as it is
    ! and is told the host's new atom order (msevb_set_host_order).  The Verlet list lives on the device: nothing to rebuild here.
    if ( new_i_mole_hydronium /= i_mole_hydronium ) then
       call evb_change_diabat_data_structure_topology( principle_diabat, i_mole_hydronium, system_data, molecule_data, atom_data, hydronium_molecule_index )
       call msevb_hip_remap( system_data, molecule_data )
    endif

  end subroutine ms_evb_calculate_total_force_energy

end module ms_evb
