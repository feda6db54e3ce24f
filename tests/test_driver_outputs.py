"""The offline driver's other output files: save_net_fluxes (radiation_save.F90:464) and
save_sw_diagnostics (:1314) with config%get_sw_mapping (radiation_config.F90:1766), through `python -m ecrad_amd.driver`
pieces on the CPU with the oracle as the backend (the driver logic is host code; the GPU path has its own tests)."""
import os

import numpy as np

from ecrad_amd.driver import get_sw_mapping, save_fluxes, save_net_fluxes, save_sw_diagnostics
from ecrad_amd.ncfile import NcFile
from helpers import make_config, rel_err, run_case


def test_net_flux_file_has_the_references_variables(tmp_path, oracle_lib):
    config = make_config("Tripleclouds", do_lw_derivatives=True, do_canopy_fluxes_sw=True, do_canopy_fluxes_lw=True)
    flux, th, _ = run_case(config, oracle_lib.backend)
    p = str(tmp_path / "net.nc")
    save_net_fluxes(p, config, th, flux, is_double_precision=True, experiment_name="unit test")
    with NcFile(p) as f:
        names = set(f._f.variables)
        want = {"pressure_hl", "flux_net_lw", "flux_dn_lw_surf", "flux_net_lw_clear", "flux_dn_lw_clear_surf", "lw_derivative",
                "canopy_flux_dn_lw_surf", "flux_net_sw", "flux_dn_sw_surf", "flux_dn_sw_toa", "flux_dn_direct_sw_surf",
                "flux_net_sw_clear", "flux_dn_sw_clear_surf", "flux_dn_direct_sw_clear_surf",
                "canopy_flux_dn_diffuse_sw_surf", "canopy_flux_dn_direct_sw_surf"}
        assert names == want
        assert np.array_equal(f.get("flux_net_sw"), (flux.sw_dn - flux.sw_up).T)
        assert np.array_equal(f.get("flux_net_lw_clear"), (flux.lw_dn_clear - flux.lw_up_clear).T)
        assert np.array_equal(f.get("flux_dn_sw_toa"), flux.sw_dn[0])
        assert np.array_equal(f.get("flux_dn_lw_surf"), flux.lw_dn[-1])
        assert f.global_attr("experiment") == "unit test"
    # and it carries the same information as the full file
    p2 = str(tmp_path / "full.nc")
    save_fluxes(p2, config, th, flux, is_double_precision=True)
    with NcFile(p) as a, NcFile(p2) as b:
        assert np.array_equal(a.get("flux_net_lw"), b.get("flux_dn_lw") - b.get("flux_up_lw"))


def test_sw_diagnostics_partition_the_surface_flux(tmp_path, oracle_lib):
    """Intervals that cover the whole shortwave spectrum must add up to the broadband surface fluxes; UV / visible / near
    infrared bounds as in the reference's documentation of sw_diag_wavelength_bound."""
    config = make_config("Tripleclouds", do_save_spectral_flux=True)
    flux, th, _ = run_case(config, oracle_lib.backend)
    bounds = [1.0e-7, 4.0e-7, 7.0e-7, 4.0e-6, 1.0e-3]
    mapping = get_sw_mapping(config, bounds)
    assert mapping.shape == (4, config.n_bands_sw)
    assert np.all(mapping >= 0.0)
    # every band's flux is shared out between the intervals (and the two discarded end intervals get next to nothing)
    assert np.abs(mapping.sum(axis=0) - 1.0).max() < 1.0e-3
    p = str(tmp_path / "sw_diag.nc")
    save_sw_diagnostics(p, config, bounds, mapping, flux, is_double_precision=True)
    with NcFile(p) as f:
        assert np.array_equal(f.get("wavelength1"), np.array(bounds[:4]))
        assert np.array_equal(f.get("wavelength2"), np.array(bounds[1:]))
        day = flux.sw_dn[0] > 0
        assert rel_err(f.get("flux_dn_sw_surf").sum(axis=1)[day], flux.sw_dn[-1][day]) < 2.0e-3
        assert rel_err(f.get("flux_dn_direct_sw_surf").sum(axis=1)[day], flux.sw_dn_direct[-1][day]) < 2.0e-3
        assert rel_err(f.get("flux_up_sw_toa").sum(axis=1)[day], flux.sw_up[0][day]) < 2.0e-3
        assert rel_err(f.get("flux_dn_sw_surf_clear").sum(axis=1)[day], flux.sw_dn_clear[-1][day]) < 2.0e-3
        for name in ("flux_up_sw_surf", "flux_dn_sw_toa", "flux_up_sw_toa_clear", "flux_up_sw_surf_clear", "flux_dn_direct_sw_surf_clear"):
            assert f.exists(name), name
        # the visible interval carries a large part of the surface flux under a high sun, the UV one little
        vis = f.get("flux_dn_sw_surf")[:, 1]
        uv = f.get("flux_dn_sw_surf")[:, 0]
        assert np.all(uv[day] < vis[day])


def test_driver_main_writes_net_fluxes_and_diagnostics(tmp_path, oracle_lib, monkeypatch):
    """The namelist route: do_save_net_fluxes and sw_diag_wavelength_bound in &radiation_driver."""
    from ecrad_amd import driver, interface
    from helpers import MERIDIAN, NAMELIST
    import re
    nam = open(NAMELIST).read()
    nam = re.sub(r"(?m)^\s*do_save_net_fluxes\s*=.*$", "", nam)        # (the test namelist sets it to false)
    nam = nam.replace("&radiation_driver", "&radiation_driver\n do_save_net_fluxes = .true.,\n sw_diag_wavelength_bound = 2.0e-7, 7.0e-7, 5.0e-6,\n"
                      f" sw_diag_file_name = \"{tmp_path}/diag.nc\",", 1)
    from helpers import DATA_DIR
    nam = re.sub(r'directory_name\s*=\s*"[^"]*"', f'directory_name = "{DATA_DIR}"', nam)
    cfg = tmp_path / "config.nam"
    cfg.write_text(nam)
    real = interface.Radiation
    monkeypatch.setattr(interface, "Radiation", lambda config, **kw: real(config, backend=oracle_lib.backend))
    out = str(tmp_path / "out.nc")
    assert driver.main([str(cfg), MERIDIAN, out]) == 0
    with NcFile(out) as f:
        assert f.exists("flux_net_sw") and not f.exists("flux_up_sw")
    with NcFile(str(tmp_path / "diag.nc")) as f:
        assert f.get("flux_dn_sw_surf").shape == (32, 2)


def test_radiative_properties_file(tmp_path, oracle_lib, monkeypatch):
    """do_save_radiative_properties (radiation_interface.F90:403-419 -> save_radiative_properties, radiation_save.F90:716-1021):
    radiation() itself dumps the arrays it passes between its stages; the reference's variable names and conditions, the
    values those of the optics entry point, the file name that of a whole-range or of a sub-range call."""
    monkeypatch.chdir(tmp_path)
    config = make_config("Tripleclouds", do_save_radiative_properties=True)
    flux, th, rad = run_case(config, oracle_lib.backend)
    assert os.path.exists("radiative_properties.nc")
    with NcFile("radiative_properties.nc") as f:
        names = set(f._f.variables)
        want = {"pressure_hl", "q_sat_liquid", "cos_solar_zenith_angle", "cloud_fraction", "overlap_param", "planck_hl", "lw_emission",
                "lw_emissivity", "od_lw", "od_lw_cloud", "ssa_lw_cloud", "asymmetry_lw_cloud", "incoming_sw", "sw_albedo",
                "sw_albedo_direct", "od_sw", "ssa_sw", "asymmetry_sw", "od_sw_cloud", "ssa_sw_cloud", "asymmetry_sw_cloud",
                "fractional_std"}       # (the cloud effective sizes are only there for SPARTACUS)
        assert names == want, names ^ want
        assert f.get("od_sw").shape == (32, 137, 32) and f.get("planck_hl").shape == (32, 138, 32) and f.get("od_lw_cloud").shape == (32, 137, 32)
        od_sw, ssa_sw, emis = f.get("od_sw"), f.get("ssa_sw"), f.get("lw_emissivity")
        assert np.all(od_sw > 0.0) and np.all((ssa_sw >= 0.0) & (ssa_sw <= 1.0)) and np.all((emis > 0.9) & (emis <= 1.0))
        assert np.array_equal(f.get("pressure_hl"), th.pressure_hl.T)
        # the incoming flux at the top of the atmosphere adds up to the solar irradiance
    # a sub-range call names its file after the range and holds that range only
    from helpers import load_meridian
    from ecrad_amd.types import Flux
    ncol, nlev, sl, th2, gas, cloud, aer = load_meridian(config)
    rad.set_gas_units(gas)
    th2.calc_saturation_wrt_liquid()
    rad.radiation(ncol, nlev, 5, 12, sl, th2, gas, cloud, aer, Flux.allocate(config, ncol, nlev))
    with NcFile("radiative_properties_0005-0012.nc") as g, NcFile("radiative_properties.nc") as f:
        assert g.get("od_lw").shape == (8, 137, 32)
        assert np.array_equal(g.get("od_lw"), f.get("od_lw")[4:12])
        assert np.array_equal(g.get("cos_solar_zenith_angle"), f.get("cos_solar_zenith_angle")[4:12])


import pytest  # noqa: E402


@pytest.mark.gpu
def test_radiative_properties_file_from_the_hip_path(tmp_path, oracle_lib, monkeypatch):
    """The same file written by radiation() on the GPU (ecrad_hip_optics) and with the oracle as the backend."""
    files = {}
    for tag, backend in (("hip", "hip"), ("oracle", oracle_lib.backend)):
        d = tmp_path / tag
        d.mkdir()
        monkeypatch.chdir(d)
        _, _, rad = run_case(make_config("McICA", do_save_radiative_properties=True, do_lw_aerosol_scattering=True), backend)
        if tag == "hip":
            rad.close()
        files[tag] = str(d / "radiative_properties.nc")
    with NcFile(files["hip"]) as a, NcFile(files["oracle"]) as b:
        assert set(a._f.variables) == set(b._f.variables) and "ssa_lw" in a._f.variables
        for name in a._f.variables:
            assert rel_err(a.get(name), b.get(name), floor_frac=1e-9) < 1e-10, name


def test_saved_inputs_read_back_as_the_inputs(tmp_path):
    """save_inputs (radiation_save.F90:1026-1320; the driver's do_save_inputs): the file must be an input file of the driver
    again -- what it reads from it is what was saved (the gases through their unit conversions: last bits)."""
    from ecrad_amd.driver import DriverConfig, read_input, save_inputs
    from helpers import MERIDIAN, NAMELIST
    config = make_config("SPARTACUS")
    dc = DriverConfig.read(NAMELIST)
    a = read_input(MERIDIAN, config, dc)
    p = str(tmp_path / "inputs.nc")
    save_inputs(p, config, a[2], a[3], a[4], a[5], a[6], lat=np.zeros(a[0]), lon=np.zeros(a[0]))
    # (the effective cloud sizes now come from the file, not from the namelist's parametrisation)
    dc2 = DriverConfig.read(NAMELIST)
    dc2.cloud_separation_scale_surface = dc2.cloud_separation_scale_toa = -1.0
    b = read_input(p, config, dc2)
    assert a[:2] == b[:2]
    for obj_a, obj_b, names in ((a[2], b[2], ("cos_sza", "skin_temperature", "sw_albedo", "lw_emissivity", "sw_albedo_direct", "iseed")),
                                (a[3], b[3], ("pressure_hl", "temperature_hl")),
                                (a[5], b[5], ("fraction", "mixing_ratio", "effective_radius", "fractional_std", "overlap_param",
                                              "inv_cloud_effective_size", "inv_inhom_effective_size")),
                                (a[6], b[6], ("mixing_ratio",))):
        for n in names:
            x, y = getattr(obj_a, n), getattr(obj_b, n)
            assert (x is None) == (y is None), n
            if x is not None:
                assert np.array_equal(np.asarray(x), np.asarray(y)), n
    assert a[2].solar_irradiance == b[2].solar_irradiance
    ga, gb = a[4], b[4]
    assert ga.is_present == gb.is_present
    from ecrad_amd.types import IVolumeMixingRatio
    for jgas in range(1, 13):
        if ga.is_present[jgas]:
            assert rel_err(gb.get(jgas, IVolumeMixingRatio), ga.get(jgas, IVolumeMixingRatio), floor_frac=1e-30) < 1e-14, jgas
    with NcFile(p) as f:
        assert {"q", "o3_mmr", "co2_vmr", "lat", "lon", "aerosol_mmr", "re_ice", "iseed"} <= set(f._f.variables)


def test_aerosol_optics_file(tmp_path):
    """aerosol_optics_type%save (the driver's do_save_aerosol_optics; the reference's test_aerosol_averaging target): the
    mapped tables under the reference's names, per g-point (ecCKD) and per band (RRTMG)."""
    from ecrad_amd.interface import setup_radiation
    from helpers import make_config_rrtmg
    for config, nsw, nlw in ((make_config("Tripleclouds"), 32, 32), (make_config_rrtmg("McICA"), 14, 16)):
        setup_radiation(config)
        p = str(tmp_path / f"aerosol_optics_{nsw}.nc")
        config.aerosol_optics.save(p)
        ao = config.aerosol_optics
        with NcFile(p) as f:
            assert {"mass_ext_sw_hydrophobic", "ssa_sw_hydrophobic", "asymmetry_sw_hydrophobic", "mass_ext_lw_hydrophobic", "ssa_lw_hydrophobic",
                    "asymmetry_lw_hydrophobic", "mass_ext_sw_hydrophilic", "ssa_sw_hydrophilic", "asymmetry_sw_hydrophilic",
                    "mass_ext_lw_hydrophilic", "ssa_lw_hydrophilic", "asymmetry_lw_hydrophilic"} == set(f._f.variables)
            assert f.get("mass_ext_sw_hydrophobic").shape == (ao.n_type_phobic, nsw)
            assert f.get("asymmetry_lw_hydrophilic").shape == (ao.n_type_philic, ao.nrh, nlw)
            assert np.array_equal(f.get("ssa_sw_hydrophilic"), ao.ssa_sw_philic)
            assert np.all((f.get("ssa_lw_hydrophobic") >= 0.0) & (f.get("ssa_lw_hydrophobic") <= 1.0))


def test_cloud_optics_files(tmp_path):
    """general_cloud_optics_type%save (the driver's do_save_cloud_optics): one file per hydrometeor type and spectrum."""
    from ecrad_amd.interface import setup_radiation
    config = make_config("Tripleclouds")
    setup_radiation(config)
    for tag, tables in (("sw", config.cloud_optics_sw), ("lw", config.cloud_optics_lw)):
        for co in tables:
            p = str(tmp_path / f"hydrometeor_optics_{tag}_{co.type_name}.nc")
            co.save(p)
            with NcFile(p) as f:
                assert f.get("mass_extinction_coefficient").shape == (co.n_effective_radius, co.n_bands)
                assert np.array_equal(f.get("single_scattering_albedo"), co.ssa)
                re = f.get("effective_radius")
                assert re[0] == co.effective_radius_0 and abs((re[1] - re[0]) / co.d_effective_radius - 1.0) < 1e-12


def test_overlap_decorrelation_length_override():
    """overlap_decorr_length_override of the driver namelist (ecrad_driver_read_input.F90:233-246): cloud%set_overlap_param
    (pinned to the reference's routine in tests/test_ifs_scheme.py) replaces the file's overlap parameter."""
    from ecrad_amd.driver import DriverConfig, read_input
    from ecrad_amd.ifs import set_overlap_param
    from helpers import MERIDIAN, NAMELIST
    config = make_config("Tripleclouds")
    dc = DriverConfig.read(NAMELIST)
    dc.overlap_decorr_length_override = 1500.0
    inp = read_input(MERIDIAN, config, dc)
    assert np.array_equal(inp[5].overlap_param, set_overlap_param(inp[3], 1500.0))
    assert np.all((inp[5].overlap_param > 0.0) & (inp[5].overlap_param < 1.0))


def test_effective_size_overrides_by_height():
    """[low|middle|high]_inv_effective_size_override (ecrad_driver_read_input.F90:305-331 -> create_inv_cloud_effective_size_eta,
    radiation_cloud.F90:524-594): one inverse cloud size per height range, split at eta = p/p_surface = 0.8 and 0.45."""
    from ecrad_amd.driver import DriverConfig, read_input
    from helpers import MERIDIAN, NAMELIST
    config = make_config("SPARTACUS")
    dc = DriverConfig.read(NAMELIST)
    dc.low_inv_effective_size_override, dc.middle_inv_effective_size_override, dc.high_inv_effective_size_override = 1.0e-3, 5.0e-4, 1.0e-4
    inp = read_input(MERIDIAN, config, dc)
    ics, p = inp[5].inv_cloud_effective_size, inp[3].pressure_hl
    assert inp[5].inv_inhom_effective_size is None and ics.shape == (137, 32)
    eta = 0.5 * (p[:-1] + p[1:]) / p[-1]
    assert np.array_equal(ics == 1.0e-3, eta > 0.8) and np.array_equal(ics == 1.0e-4, eta <= 0.45)
    assert set(np.unique(ics)) == {1.0e-3, 5.0e-4, 1.0e-4}


def test_band_wise_aerosol_file_is_read_as_it_is():
    """use_general_aerosol_optics = false (aerosol_optics_type%setup, radiation_aerosol_optics_data.F90:157-315): the
    properties are already in the RRTMG bands -- no averaging, the arrays of the file."""
    from ecrad_amd.interface import setup_radiation
    from helpers import DATA_DIR, make_config_rrtmg
    config = make_config_rrtmg("McICA", use_general_aerosol_optics=False)
    setup_radiation(config)
    ao = config.aerosol_optics
    assert (ao.n_bands_sw, ao.n_bands_lw, ao.n_type_phobic, ao.n_type_philic, ao.nrh) == (14, 16, 14, 10, 12)
    with NcFile(os.path.join(DATA_DIR, "aerosol_ifs_rrtm_46R1_with_NI_AM.nc")) as f:
        assert np.array_equal(ao.mass_ext_sw_phobic, f.get("mass_ext_sw_hydrophobic"))
        assert np.array_equal(ao.g_lw_philic, f.get("asymmetry_lw_hydrophilic"))
        assert np.array_equal(ao.rh_lower, f.get("relative_humidity1"))
    # and the ecCKD models cannot use it
    import pytest
    from ecrad_amd.config import ConfigError
    with pytest.raises(ConfigError):
        setup_radiation(make_config("Tripleclouds", use_general_aerosol_optics=False))


@pytest.mark.parametrize("with_inhom", [False, True])
def test_cloud_size_from_effective_separation_in_the_input_file(tmp_path, with_inhom):
    """Case (4) of driver/ecrad_driver_read_input.F90:290-470: the input file holds inv_cloud_effective_separation [and
    inv_inhom_effective_separation] instead of the effective sizes.  The reference's own executable reads such a file and
    dumps what it hands to radiation() (do_save_inputs -> inputs.nc); the Python driver derives the same
    inv_cloud_effective_size / inv_inhom_effective_size from it, including effective_size_scaling."""
    import re
    import subprocess
    from scipy.io import netcdf_file
    from ecrad_amd.driver import DriverConfig, read_input
    from helpers import MERIDIAN, NAMELIST
    from test_fortran_dropin import write_namelist
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "reference", "ecrad_ref")
    if not os.path.exists(ref):
        pytest.skip("tests/_build/reference/ecrad_ref has not been built (tools/build_dropin.py --reference)")
    # the meridian file with the separations in place of the sizes
    src = netcdf_file(MERIDIAN, "r", mmap=False)
    inp = str(tmp_path / "separation.nc")
    dst = netcdf_file(inp, "w", version=1)
    for d, n in src.dimensions.items():
        dst.createDimension(d, n)
    rng = np.random.default_rng(5)
    shape = None
    for name, v in src.variables.items():
        if name in ("inv_cloud_effective_size", "inv_inhom_effective_size"):
            shape = (v.dimensions, v.shape)
            continue
        o = dst.createVariable(name, v.data.dtype.char if v.data.dtype.kind != "f" else v.data.dtype.char, v.dimensions)
        if v.shape == ():
            o.data[...] = v.data
        else:
            o[:] = v[:]
    assert shape is not None
    for name in ["inv_cloud_effective_separation"] + (["inv_inhom_effective_separation"] if with_inhom else []):
        o = dst.createVariable(name, "d", shape[0])
        o[:] = 1.0 / rng.uniform(500.0, 20000.0, size=shape[1])
    dst.close()
    src.close()
    nam = str(tmp_path / "config.nam")
    write_namelist(nam, {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "do_3d_effects": "true"})
    text = open(nam).read()
    text = re.sub(r"do_save_inputs\s*=\s*false", "do_save_inputs = true", text)
    text = re.sub(r"cloud_separation_scale_toa\s*=\s*[0-9.]+", "cloud_separation_scale_toa = -1.0", text)
    text = re.sub(r"cloud_separation_scale_surface\s*=\s*[0-9.]+", "cloud_separation_scale_surface = -1.0", text)
    text = text.replace("&radiation_driver\n", "&radiation_driver\neffective_size_scaling = 1.7,\n", 1)
    open(nam, "w").write(text)
    p = subprocess.run(f"ulimit -s unlimited; exec {ref} {nam} {inp} {tmp_path / 'out.nc'}", shell=True, capture_output=True, text=True,
                       cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="2", OMP_STACKSIZE="1G"), timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "inv_cloud_effective_separation" in p.stdout + p.stderr
    config = make_config("SPARTACUS", do_3d_effects=True)
    dc = DriverConfig.read(nam)
    assert dc.effective_size_scaling == 1.7 and dc.cloud_separation_scale_toa < 0.0
    cloud = read_input(inp, config, dc)[5]
    with NcFile(str(tmp_path / "inputs.nc")) as f:
        for name in ("inv_cloud_effective_size", "inv_inhom_effective_size"):
            want = f.get(name)                                  # (column, level) as the file has it
            got = np.asarray(getattr(cloud, name)).T            # the host's arrays are (level, column)
            assert want.shape == got.shape, name
            assert want.max() > 0.0
            assert rel_err(got, want, floor_frac=1e-30) < 1e-6, name      # (inputs.nc is written in single precision)
