"""draco-sharp_amd: MI355X-native Draco mesh-decode path behind draco-sharp's
DracoDecoder / Mesh / PointCloud surface.  The decode product lives in csrc/
(HIP kernels + C-ABI, see include/draco_mi355x.h); this package is the thin
host-side mirror used by the tests and bench harness."""
__version__ = "0.1.0"

from .decoder import (Batch, Context, DataBuffer, DeviceException, Draco, DracoDecoder, DracoHeader, DracoMetadata,  # noqa: E402,F401
                      InvalidDataException, Mesh, MetadataElement, PointAttribute, PointCloud, Pool, PoolJob, parse_metadata, pool_plan)
from .encoder import Config, DracoEncoder, MeshData  # noqa: E402,F401
