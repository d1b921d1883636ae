import sys, os
sys.path.insert(0, os.getcwd())
import torch
from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex
stream = torch.cuda.current_stream().cuda_stream
polys = DeviceGeoArray.upload(synth.star_polygons(1000, 64), stream=stream)
os.environ["GPK_DEBUG_INDEX"]="1"
index = SpatialIndex.from_device(polys, stream=stream)
print(index.describe())
