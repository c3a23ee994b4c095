"""Static geometry: icosphere vs the reference's own module (golden fixture), the
reference's mesh / connectivity tests restated (common/icosahedral_mesh_test.py:36-131,
common/grid_mesh_connectivity_test.py:24-71), and the survey's measured nano counts."""
import os

import numpy as np
import pytest

from gencast_flax_nnx_amd import geometry as g
from gencast_flax_nnx_amd import geometry
from oracle import gencast_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "icosphere.npz"))


def _mesh_spec(splits):
  nv, nf = 12, 20
  for _ in range(splits):
    nv += nf * 3 // 2
    nf *= 4
  return nv, nf


def _assert_valid_mesh(mesh, nv, nf):
  assert mesh.vertices.shape == (nv, 3) and mesh.faces.shape == (nf, 3)
  np.testing.assert_allclose(np.linalg.norm(mesh.vertices, axis=-1), 1.0, rtol=1e-6)
  v, f = mesh.vertices, mesh.faces
  orient = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 1]])
  orient /= np.linalg.norm(orient, axis=-1, keepdims=True)
  centers = v[f].mean(1)
  centers /= np.linalg.norm(centers, axis=-1, keepdims=True)
  np.testing.assert_allclose(np.einsum("ik,ik->i", orient, centers), 1.0, atol=6e-4)


@pytest.mark.parametrize("splits", range(6))
def test_icosphere_matches_reference_bit_for_bit(splits):
  m = g.get_last_triangular_mesh_for_sphere(splits)
  assert m.vertices.dtype == np.float32 and m.faces.dtype == np.int32
  np.testing.assert_array_equal(m.vertices, GOLD[f"vertices_{splits}"])
  np.testing.assert_array_equal(m.faces, GOLD[f"faces_{splits}"])
  s, r = g.faces_to_edges(m.faces)
  np.testing.assert_array_equal(s, GOLD[f"senders_{splits}"])
  np.testing.assert_array_equal(r, GOLD[f"receivers_{splits}"])


def test_icosahedron():
  _assert_valid_mesh(g.icosahedron(), 12, 20)


@pytest.mark.parametrize("splits", range(5))
def test_hierarchy_of_meshes(splits):
  meshes = g.get_hierarchy_of_triangular_meshes_for_sphere(splits)
  prev = None
  for i, mesh in enumerate(meshes):
    _assert_valid_mesh(mesh, *_mesh_spec(i))
    if prev is not None:
      np.testing.assert_array_equal(mesh.vertices[:prev.shape[0]], prev)
    prev = mesh.vertices


def test_faces_to_edges_order():
  faces = np.array([[0, 1, 2], [3, 4, 5]])
  expected = np.array([[0, 1], [3, 4], [1, 2], [4, 5], [2, 0], [5, 3]])
  s, r = g.faces_to_edges(faces)
  np.testing.assert_array_equal(s, expected[:, 0])
  np.testing.assert_array_equal(r, expected[:, 1])


def test_grid_lat_lon_to_coordinates_kat():
  lat = np.array([-45.0, 0.0, 45])
  lon = np.array([0.0, 90.0, 180.0, 270.0])
  i2 = 1 / np.sqrt(2)
  expected = np.array([
      [[i2, 0, -i2], [0, i2, -i2], [-i2, 0, -i2], [0, -i2, -i2]],
      [[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]],
      [[i2, 0, i2], [0, i2, i2], [-i2, 0, i2], [0, -i2, i2]]], dtype=np.float64)
  np.testing.assert_allclose(g.grid_lat_lon_to_coordinates(lat, lon), expected, atol=1e-15)


def _smoke_grid():
  return np.linspace(-75, 75, 6), np.arange(12) * 30.0, g.get_last_triangular_mesh_for_sphere(3)


def test_radius_query_matches_brute_force():
  lat, lon, mesh = _smoke_grid()
  gi, mi = g.radius_query_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh, radius=0.2)
  pos = g.grid_lat_lon_to_coordinates(lat, lon).reshape(-1, 3)
  d = np.linalg.norm(pos[:, None, :] - mesh.vertices[None].astype(np.float64), axis=-1)
  bg, bm = np.nonzero(d <= 0.2)
  np.testing.assert_array_equal(gi, bg)
  np.testing.assert_array_equal(mi, bm)
  assert len(gi) > 0


def test_in_mesh_triangle_is_closest_face():
  lat, lon, mesh = _smoke_grid()
  gi, mi = g.in_mesh_triangle_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh)
  n = lat.size * lon.size
  np.testing.assert_array_equal(gi, np.repeat(np.arange(n), 3))
  pos = g.grid_lat_lon_to_coordinates(lat, lon).reshape(-1, 3)
  v = mesh.vertices.astype(np.float64)
  chosen = mi.reshape(n, 3)
  # the chosen triple is a face of the mesh, and no face is closer (brute force over all faces)
  face_set = {tuple(f) for f in mesh.faces.tolist()}
  best = np.full(n, np.inf)
  for f in mesh.faces:
    d = g._closest_point_sqdist_on_triangles(pos, np.repeat(v[f[0]][None], n, 0),
                                             np.repeat(v[f[1]][None], n, 0),
                                             np.repeat(v[f[2]][None], n, 0))
    best = np.minimum(best, d)
  d_chosen = g._closest_point_sqdist_on_triangles(pos, v[chosen[:, 0]], v[chosen[:, 1]], v[chosen[:, 2]])
  assert all(tuple(c) in face_set for c in chosen.tolist())
  np.testing.assert_allclose(d_chosen, best, atol=1e-12)


def test_edge_features_closed_form_equals_scipy_rotation():
  lat, lon, mesh = _smoke_grid()
  g_lat, g_lon = g.grid_nodes_lat_lon(lat.astype(np.float32), lon.astype(np.float32))
  m_lat, m_lon = g.mesh_nodes_lat_lon(mesh)
  gi, mi = g.radius_query_indices(grid_latitude=lat, grid_longitude=lon, mesh=mesh, radius=0.3)
  ours = g.bipartite_edge_structural_features(
      senders_lat=g_lat, senders_lon=g_lon, receivers_lat=m_lat, receivers_lon=m_lon,
      senders=gi, receivers=mi)
  ref = O.bipartite_edge_features_scipy(g_lat, g_lon, m_lat, m_lon, gi, mi)
  np.testing.assert_allclose(ours, ref, atol=1e-12)
  assert ours.dtype == np.float64 and ours.shape == (len(gi), 4)
  np.testing.assert_allclose(ours[:, 0].max(), 1.0)
  # receiver frame: the receiver sits at (1,0,0), so |d|^2 = dx^2+dy^2+dz^2
  np.testing.assert_allclose(ours[:, 0] ** 2, (ours[:, 1:] ** 2).sum(-1), atol=1e-12)


def test_node_features():
  lat = np.array([90.0, 0.0, -90.0], dtype=np.float32)
  lon = np.array([0.0, 90.0, 180.0], dtype=np.float32)
  f = g.node_structural_features(lat, lon)
  assert f.dtype == np.float32
  np.testing.assert_allclose(f, [[1, 1, 0], [0, 0, 1], [-1, -1, 0]], atol=1e-6)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_khop_equals_matrix_power_pattern(k):
  mesh = g.get_last_triangular_mesh_for_sphere(2)
  s, r = g.faces_to_edges(mesh.faces)
  rowptr, cols = g.khop_neighbourhood_csr(mesh.vertices.shape[0], s, r, k)
  ref = O.khop_mask(mesh.vertices.shape[0], s, r, k)
  ref = (ref != 0).tocsr()
  ref.sort_indices()
  np.testing.assert_array_equal(rowptr, ref.indptr)
  np.testing.assert_array_equal(cols, ref.indices)


def test_nano_graph_counts_match_survey():
  """SURVEY.md 8d [measured]: E1=16830, E2=31536, nnz=542922, in-degree 3..218,
  max edge 0.082604, block size 649 under the reference's RCM order."""
  lat = np.arange(-90, 90.01, 2.5)
  lon = np.arange(0, 360, 2.5)
  gr = g.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=4, attention_k_hop=8)
  assert (gr.num_grid_nodes, gr.num_mesh_nodes) == (10512, 2562)
  assert len(gr.g2m_senders) == 16830 and len(gr.m2g_senders) == 31536
  assert len(gr.khop_cols) == 542922
  deg = np.bincount(gr.g2m_receivers, minlength=2562)
  assert (deg.min(), deg.max()) == (3, 218)
  np.testing.assert_array_equal(gr.m2g_receivers, np.repeat(np.arange(10512), 3))
  mesh = g.get_last_triangular_mesh_for_sphere(4)
  np.testing.assert_allclose(g.max_edge_distance(mesh), 0.082604, atol=5e-7)
  import scipy.sparse
  perm = O.rcm_permutation(2562, gr.mesh_senders, gr.mesh_receivers)
  mask = scipy.sparse.csr_matrix((np.ones(len(gr.khop_cols), np.int32), gr.khop_cols, gr.khop_rowptr),
                                 shape=(2562, 2562))
  assert O.get_mask_block_size(mask[perm][:, perm].tocsr()) == 649


def test_reference_graph_injection_reproduces_the_built_graph(tmp_path):
  """`graph_from_reference_arrays` on the arrays the INTEGRATION.md snippet dumps from the reference: fed
  with this build's own arrays under a mesh RENUMBERING (the reference numbers mesh nodes by RCM), it must
  give the same graph up to that renumbering -- edges, structural features and k-hop sets."""
  lat, lon = np.linspace(-90, 90, 13), np.arange(24) * 15.0
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=2, attention_k_hop=2)
  M = gr.num_mesh_nodes
  perm = np.random.default_rng(0).permutation(M)             # new id -> old id
  inv = np.empty(M, np.int64)
  inv[perm] = np.arange(M)
  mesh = geometry.get_last_triangular_mesh_for_sphere(2)
  m_lat, m_lon = geometry.mesh_nodes_lat_lon(mesh)
  path = str(tmp_path / "ref_graph.npz")
  np.savez(path, mesh_nodes_lat=m_lat[perm], mesh_nodes_lon=m_lon[perm], g2m_senders=gr.g2m_senders,
           g2m_receivers=inv[gr.g2m_receivers], m2g_senders=inv[gr.m2g_senders], m2g_receivers=gr.m2g_receivers,
           mesh_senders=inv[gr.mesh_senders], mesh_receivers=inv[gr.mesh_receivers])
  g2 = geometry.load_reference_graph(path, grid_lat=lat, grid_lon=lon, attention_k_hop=2)
  assert (g2.num_grid_nodes, g2.num_mesh_nodes) == (gr.num_grid_nodes, M)
  np.testing.assert_array_equal(perm[g2.g2m_receivers], gr.g2m_receivers)
  np.testing.assert_array_equal(perm[g2.m2g_senders], gr.m2g_senders)
  np.testing.assert_allclose(g2.g2m_edge_struct, gr.g2m_edge_struct, atol=2e-6)
  np.testing.assert_allclose(g2.m2g_edge_struct, gr.m2g_edge_struct, atol=2e-6)
  np.testing.assert_allclose(g2.mesh_struct, gr.mesh_struct[perm], atol=2e-6)
  np.testing.assert_allclose(g2.mesh_xyz, gr.mesh_xyz[perm], atol=2e-6)
  for new in (0, 7, M - 1):
    a = sorted(perm[g2.khop_cols[g2.khop_rowptr[new]:g2.khop_rowptr[new + 1]]].tolist())
    old = perm[new]
    assert a == sorted(gr.khop_cols[gr.khop_rowptr[old]:gr.khop_rowptr[old + 1]].tolist())
  with pytest.raises(ValueError, match="out of range"):
    geometry.graph_from_reference_arrays(
        grid_lat=lat, grid_lon=lon, mesh_nodes_lat=m_lat, mesh_nodes_lon=m_lon, g2m_senders=gr.g2m_senders,
        g2m_receivers=gr.g2m_receivers + M, m2g_senders=gr.m2g_senders, m2g_receivers=gr.m2g_receivers,
        mesh_senders=gr.mesh_senders, mesh_receivers=gr.mesh_receivers, attention_k_hop=2)


def test_m2g_tie_nodes_are_counted():
  """174 of the 10 512 nano grid points (444 of 65 160 at 1 deg / mesh 5) sit on a shared
  mesh edge or vertex, where the reference's trimesh query may pick another face than this build's
  lowest-face-index rule -- the reason `Denoiser(graph=...)` exists."""
  lat, lon = np.arange(-90, 90 + 1e-9, 2.5), np.arange(0, 360, 2.5)
  mesh = geometry.get_last_triangular_mesh_for_sphere(4)
  assert geometry.count_m2g_ties(grid_latitude=lat, grid_longitude=lon, mesh=mesh) == 174
  lat1, lon1 = np.arange(-90, 90 + 1e-9, 1.0), np.arange(0, 360, 1.0)
  assert geometry.count_m2g_ties(grid_latitude=lat1, grid_longitude=lon1,
                                 mesh=geometry.get_last_triangular_mesh_for_sphere(5)) == 444
