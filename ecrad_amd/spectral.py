"""Host-side spectral mapping (setup time only; never on the per-call path).

Restates the parts of radiation/radiation_spectral_definition.F90 that setup_radiation needs for
ecCKD gas models: reading the spectral definition out of a ckd-definition file (:91-137), the
wavenumber -> g-point/band mapping matrix used for cloud and aerosol optics (calc_mapping :222-493)
and the albedo/emissivity-interval mapping (calc_mapping_from_bands :515-760).  All arithmetic is
float64, operating on tables widened from the float32 the files hold.
"""
from __future__ import annotations

import numpy as np

# radiation_spectral_definition.F90:27-28
SOLAR_REFERENCE_TEMPERATURE = 5777.0
TERRESTRIAL_REFERENCE_TEMPERATURE = 273.15

# radiation_constants.F90:24-33
SPEED_OF_LIGHT = 299792458.0
BOLTZMANN_CONSTANT = 1.380648813e-23
PLANCK_CONSTANT = 6.6260695729e-34


def planck_function_wavenumber(wavenumber, temperature):
    """radiation_spectral_definition.F90:1090-1115 (W m-2 (cm-1)-1)."""
    wavenumber = np.asarray(wavenumber, dtype=np.float64)
    if temperature > 0.0:
        freq = 100.0 * SPEED_OF_LIGHT * wavenumber
        with np.errstate(over="ignore"):
            planck_fn_freq = 2.0 * PLANCK_CONSTANT * freq ** 3 / (
                SPEED_OF_LIGHT ** 2 * (np.exp(PLANCK_CONSTANT * freq
                                              / (BOLTZMANN_CONSTANT * temperature)) - 1.0))
        return planck_fn_freq * 100.0 * SPEED_OF_LIGHT
    return np.ones_like(wavenumber)


class SpectralDefinition:
    """spectral_definition_type (radiation_spectral_definition.F90:34-88)."""

    def __init__(self):
        self.nwav = 0
        self.ng = 0
        self.nband = 0
        self.wavenumber1 = None
        self.wavenumber2 = None
        self.gpoint_fraction = None      # numpy (ng, nwav)  == Fortran (nwav, ng)
        self.reference_temperature = -1.0
        self.solar_spectral_irradiance = None
        self.solar_irradiance = None
        self.wavenumber1_band = None
        self.wavenumber2_band = None
        self.i_band_number = None        # 1-based, like the reference after "+1"

    @classmethod
    def read(cls, nc) -> "SpectralDefinition":
        """read_spectral_definition (:91-137)."""
        s = cls()
        s.wavenumber1 = nc.get("wavenumber1")
        s.wavenumber2 = nc.get("wavenumber2")
        s.gpoint_fraction = nc.get("gpoint_fraction")
        s.wavenumber1_band = nc.get("wavenumber1_band")
        s.wavenumber2_band = nc.get("wavenumber2_band")
        s.i_band_number = nc.get("band_number").astype(np.int64) + 1
        if nc.exists("solar_spectral_irradiance"):
            s.solar_spectral_irradiance = nc.get("solar_spectral_irradiance")
        if nc.exists("solar_irradiance"):
            s.reference_temperature = SOLAR_REFERENCE_TEMPERATURE
            s.solar_irradiance = nc.get("solar_irradiance")
        else:
            s.reference_temperature = TERRESTRIAL_REFERENCE_TEMPERATURE
        s.nwav = s.wavenumber1.size
        s.ng = s.gpoint_fraction.shape[0]
        s.nband = s.wavenumber1_band.size
        return s

    @classmethod
    def bands_only(cls, reference_temperature, wavenumber1, wavenumber2) -> "SpectralDefinition":
        """allocate_bands_only (radiation_spectral_definition.F90:140-165): a definition that knows its bands
        but not how g-points map to wavenumber (RRTMG)."""
        s = cls()
        s.wavenumber1_band = np.asarray(wavenumber1, dtype=np.float64)
        s.wavenumber2_band = np.asarray(wavenumber2, dtype=np.float64)
        s.nband = s.wavenumber1_band.size
        s.reference_temperature = float(reference_temperature)
        s.ng = 0
        return s

    # -- find_wavenumber (:170-186): 1-based index, 0 if outside ---------------------------------
    def find(self, wavenumber: float) -> int:
        if wavenumber < self.wavenumber1[0] or wavenumber > self.wavenumber2[self.nwav - 1]:
            return 0
        i = 1
        while wavenumber > self.wavenumber2[i - 1] and i < self.nwav:
            i += 1
        return i

    def _weight_source(self):
        if self.solar_spectral_irradiance is not None:
            return self.solar_spectral_irradiance.copy()
        return planck_function_wavenumber(0.5 * (self.wavenumber1 + self.wavenumber2),
                                          self.reference_temperature)

    # -- calc_mapping (:222-493) -------------------------------------------------------------------
    def calc_mapping(self, wavenumber, weighting_temperature=None, use_bands=False):
        """Return mapping with shape (nout, nwav_in) so that y = mapping @ x."""
        wavenumber = np.asarray(wavenumber, dtype=np.float64)
        nwav = wavenumber.size
        if use_bands:
            mapping = np.zeros((self.nband, nwav))
            if weighting_temperature is not None:
                if weighting_temperature > 0.0:
                    planck_weight = planck_function_wavenumber(wavenumber, weighting_temperature)
                else:
                    planck_weight = np.ones(nwav)
            else:
                planck_weight = planck_function_wavenumber(wavenumber, self.reference_temperature)
            for jband in range(self.nband):
                w1b, w2b = self.wavenumber1_band[jband], self.wavenumber2_band[jband]
                weight = np.zeros(nwav)
                for jwav in range(nwav):
                    if w1b <= wavenumber[jwav] <= w2b:
                        if jwav > 0:
                            wavenum1 = max(w1b, 0.5 * (wavenumber[jwav - 1] + wavenumber[jwav]))
                        else:
                            wavenum1 = w1b
                        if jwav < nwav - 1:
                            wavenum2 = min(w2b, 0.5 * (wavenumber[jwav] + wavenumber[jwav + 1]))
                        else:
                            wavenum2 = w2b
                        weight[jwav] = (wavenum2 - wavenum1) * planck_weight[jwav]
                if weight.sum() <= 0.0:
                    if wavenumber[0] >= w2b:
                        weight[0] = 1.0
                    elif wavenumber[nwav - 1] <= w1b:
                        weight[nwav - 1] = 1.0
                    else:
                        iwav = 1
                        while wavenumber[iwav] < w2b:
                            iwav += 1
                        mid = 0.5 * (w2b + w1b)
                        weight[iwav - 1] = planck_weight[iwav - 1] * (wavenumber[iwav] - mid)
                        weight[iwav] = planck_weight[iwav] * (-wavenumber[iwav - 1] + mid)
                mapping[jband, :] = weight / weight.sum()
            return mapping

        if self.ng == 0:
            raise ValueError("requested cloud/aerosol mapping per g-point but only available per band")
        mapping = np.zeros((self.ng, nwav))
        planck_weight = self._weight_source()
        wn1, wn2 = self.wavenumber1, self.wavenumber2
        for jwav in range(nwav):
            weight = np.zeros(self.nwav)
            wavenum1 = wavenumber[jwav]
            isd1 = self.find(wavenum1)
            if isd1 < 1:
                continue
            i1 = isd1 - 1
            if jwav > 0:
                wavenum0 = wavenumber[jwav - 1]
                isd0 = self.find(wavenum0)
                i0 = isd0 - 1
                if isd0 == isd1:
                    weight[i0] = 0.5 * (wavenum1 - wavenum0) / (wn2[i0] - wn1[i0])
                else:
                    if isd0 >= 1:
                        weight[i0] = 0.5 * (wn2[i0] - wavenum0) ** 2 / (
                            (wn2[i0] - wn1[i0]) * (wavenum1 - wavenum0))
                    weight[i1] = 0.5 * (1.0 + (wn1[i1] - wavenum1) / (wavenum1 - wavenum0)) \
                        * (wavenum1 - wn1[i1]) / (wn2[i1] - wn1[i1])
                    if isd1 - isd0 > 1:
                        for isd in range(isd0 + 1, isd1):
                            i = isd - 1
                            weight[i] = 0.5 * (wn1[i] + wn2[i] - 2.0 * wavenum0) / (wavenum1 - wavenum0)
            else:
                weight[0:i1] = 1.0
                weight[i1] = (wavenum1 - wn1[i1]) / (wn2[i1] - wn1[i1])
            if jwav < nwav - 1:
                wavenum2 = wavenumber[jwav + 1]
                isd2 = self.find(wavenum2)
                i2 = isd2 - 1
                if isd1 == isd2:
                    weight[i1] += 0.5 * (wavenum2 - wavenum1) / (wn2[i1] - wn1[i1])
                else:
                    if 1 <= isd2 <= self.nwav:
                        weight[i2] += 0.5 * (wavenum2 - wn1[i2]) ** 2 / (
                            (wn2[i2] - wn1[i2]) * (wavenum2 - wavenum1))
                    weight[i1] += 0.5 * (1.0 + (wavenum2 - wn2[i1]) / (wavenum2 - wavenum1)) \
                        * (wn2[i1] - wavenum1) / (wn2[i1] - wn1[i1])
                    if isd2 - isd1 > 1:
                        for isd in range(isd1 + 1, isd2):
                            i = isd - 1
                            weight[i] += 0.5 * (2.0 * wavenum2 - wn1[i] - wn2[i]) / (wavenum2 - wavenum1)
            else:
                weight[i1 + 1:self.nwav] = 1.0
                weight[i1] = (wn2[i1] - wavenum1) / (wn2[i1] - wn1[i1])
            weight = weight * planck_weight
            mapping[:, jwav] = self.gpoint_fraction @ weight
        mapping *= (1.0 / mapping.sum(axis=1))[:, None]
        return mapping

    # -- calc_mapping_from_bands (:515-760), without use_fluxes --------------------------------------
    def calc_mapping_from_bands(self, wavelength_bound, i_intervals, use_bands=False, use_fluxes=False):
        """Return mapping as numpy (nout, ninput) == Fortran mapping(ninput, nout)
        (radiation_spectral_definition.F90:515-811).

        ``i_intervals`` is 1-based as in the namelist (i_sw_albedo_index / i_lw_emiss_index).  With ``use_fluxes`` the
        matrix works the other way round (:504-507): applied to fluxes per band or g-point it returns the fluxes in
        the user's intervals, each entry being the fraction of the band (g-point) that lies in the interval.
        """
        wavelength_bound = np.asarray(wavelength_bound, dtype=np.float64)
        i_intervals = np.asarray(i_intervals, dtype=np.int64)
        ninterval = i_intervals.size
        ninput = int(i_intervals.max())
        for jint in range(1, ninterval - 1):
            if wavelength_bound[jint] <= wavelength_bound[jint - 1]:
                raise ValueError("wavelength bounds must be monotonically increasing")
        if use_bands:
            mapping = np.zeros((self.nband, ninput))
            denom = np.zeros((self.nband, ninput))
            weight_sample = np.array([0.5, 1.0, 1.0, 1.0, 0.5])
            for jband in range(self.nband):
                for jint in range(ninterval):
                    wn2b = self.wavenumber2_band[jband] if jint == 0 else \
                        min(self.wavenumber2_band[jband], 0.01 / wavelength_bound[jint - 1])
                    wn1b = self.wavenumber1_band[jband] if jint == ninterval - 1 else \
                        max(self.wavenumber1_band[jband], 0.01 / wavelength_bound[jint])
                    if wn2b > wn1b:
                        sample = wn1b + np.arange(5) * (wn2b - wn1b) / 4.0
                        planck_sample = planck_function_wavenumber(sample, self.reference_temperature)
                        mapping[jband, i_intervals[jint] - 1] += \
                            np.sum(planck_sample * weight_sample) * (wn2b - wn1b)
                        if use_fluxes:      # the same integral over the whole band (:655-664)
                            w1, w2 = self.wavenumber1_band[jband], self.wavenumber2_band[jband]
                            sample = w1 + np.arange(5) * (w2 - w1) / 4.0
                            planck_sample = planck_function_wavenumber(sample, self.reference_temperature)
                            denom[jband, i_intervals[jint] - 1] += np.sum(planck_sample * weight_sample) * (w2 - w1)
            if use_fluxes:
                return mapping / np.maximum(1.0e-12, denom)
        else:
            if self.ng == 0:
                raise ValueError("requested surface mapping per g-point but only available per band")
            mapping = np.zeros((self.ng, ninput))
            planck = self._weight_source()
            for jint in range(ninterval):
                for jwav in range(self.nwav):
                    wn2b = self.wavenumber2[jwav] if jint == 0 else \
                        min(self.wavenumber2[jwav], 0.01 / wavelength_bound[jint - 1])
                    wn1b = self.wavenumber1[jwav] if jint == ninterval - 1 else \
                        max(self.wavenumber1[jwav], 0.01 / wavelength_bound[jint])
                    if wn2b > wn1b:
                        mapping[:, i_intervals[jint] - 1] += self.gpoint_fraction[:, jwav] * (
                            planck[jwav] * (wn2b - wn1b) / (self.wavenumber2[jwav] - self.wavenumber1[jwav]))
            if use_fluxes:          # :784-788
                return mapping / (self.gpoint_fraction @ planck)[:, None]
        mapping *= (1.0 / mapping.sum(axis=1))[:, None]
        return mapping
