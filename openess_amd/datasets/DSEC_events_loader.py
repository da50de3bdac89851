"""DSECEvents (datasets/DSEC_events_loader.py:6-67): the DSEC HDF5 reader needs h5py / hdf5plugin, which this
image does not ship.  The arithmetic that follows the read -- rectification, the 20 sub-window
voxel grids, the crop -- is `hip.voxelize_dsec_raw`; a native HDF5 event reader is SURVEY.md 8f row 2."""


class DSECEvents:
    def __init__(self, *a, **k):
        raise ImportError("DSEC HDF5 reading needs h5py + hdf5plugin (not installed here). Feed raw event columns to "
                          "openess_amd.hip.voxelize_dsec_raw, or use dataset_path: 'synthetic'.")

    @classmethod
    def build_from_settings(cls, s):
        return cls(), cls()
