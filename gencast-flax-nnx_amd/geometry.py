"""Init-time static geometry for the GenCast denoiser (CPU, numpy; runs once).

Builds everything `gc_set_graph` (include/gencast_hip.h) needs:

* the icosphere (reference: common/icosahedral_mesh.py:59-211,259-285),
* grid->mesh radius-query edges (common/grid_mesh_connectivity.py:40-86),
* mesh->grid containing-triangle edges (common/grid_mesh_connectivity.py:89-133),
* structural node / edge features (common/model_utils.py:24-142,205-361,364-591),
* the k-hop attention neighbourhood as CSR (gencast/transformer.py:21-47,
  gencast/sparse_transformer.py:555).

The reference renumbers mesh nodes with scipy's RCM so that its dense
tri-block-diagonal attention applies (gencast/denoiser.py:849-867).  Node
numbering is internal (weights are shared across nodes), so this build keeps the
icosphere numbering here and lets the HIP library pick its own locality order
(spatially compact 32-query tiles, see csrc/gc_graph.cpp).
"""
from __future__ import annotations

import dataclasses
from typing import Tuple

import numpy as np
import scipy.sparse
import scipy.spatial


@dataclasses.dataclass(frozen=True)
class TriangularMesh:
  """vertices [V,3] float32 (unit norm), faces [F,3] int32 (CCW from outside)."""
  vertices: np.ndarray
  faces: np.ndarray


# ----------------------------------------------------------------------------
# Icosphere
# ----------------------------------------------------------------------------

_ICOSAHEDRON_FACES = np.array(
    [(0, 1, 2), (0, 6, 1), (8, 0, 2), (8, 4, 0), (3, 8, 2), (3, 2, 7),
     (7, 2, 1), (0, 4, 6), (4, 11, 6), (6, 11, 5), (1, 5, 7), (4, 10, 11),
     (4, 8, 10), (10, 8, 3), (10, 3, 9), (11, 10, 9), (11, 9, 5), (5, 9, 7),
     (9, 3, 7), (1, 6, 5)], dtype=np.int32)


def icosahedron() -> TriangularMesh:
  """Regular icosahedron on the unit sphere, one face parallel to the xy plane.

  Same vertex order, face table and orientation as the reference
  (common/icosahedral_mesh.py:93-170): vertices are the cyclic permutations of
  (+-1, +-phi, 0)/|(1,phi)|, rotated about y by half the supplement of the
  dihedral angle.
  """
  phi = (1 + np.sqrt(5)) / 2
  verts = []
  for c1 in (1.0, -1.0):
    for c2 in (phi, -phi):
      verts.append((c1, c2, 0.0))
      verts.append((0.0, c1, c2))
      verts.append((c2, 0.0, c1))
  v = np.array(verts, dtype=np.float32)
  v /= np.linalg.norm([1.0, phi])
  angle = (np.pi - 2 * np.arcsin(phi / np.sqrt(3))) / 2
  c, s = np.cos(angle), np.sin(angle)
  # Row-vector convention: v @ R_y(angle).
  rot = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=np.float64)
  v = np.dot(v, rot)
  return TriangularMesh(v.astype(np.float32), _ICOSAHEDRON_FACES.copy())


def _split_faces(mesh: TriangularMesh) -> TriangularMesh:
  """One 1->4 triangle split; new vertices are edge midpoints pushed to the sphere.

  Vertex numbering follows first appearance while walking faces in order and,
  inside a face, edges (v1,v2), (v2,v3), (v3,v1) -- the order the reference's
  hash-table builder produces (common/icosahedral_mesh.py:173-256).
  """
  v = mesh.vertices
  f = mesh.faces.astype(np.int64)
  nv = v.shape[0]
  # half-edges in creation order: face-major, edge-minor
  he = np.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=1).reshape(-1, 2)
  lo = he.min(axis=1)
  hi = he.max(axis=1)
  key = lo * nv + hi
  uniq, first_idx, inverse = np.unique(key, return_index=True, return_inverse=True)
  order = np.argsort(first_idx, kind="stable")      # unique ids by first appearance
  rank = np.empty_like(order)
  rank[order] = np.arange(order.size)
  child_of_halfedge = nv + rank[inverse]               # [3F]
  parents = he[first_idx[order]]                       # [n_new, 2] as first written
  new_v = np.empty((order.size, 3), dtype=np.float32)
  for i in range(order.size):                          # same f32 ops as the reference
    p = v[parents[i]].mean(0)
    p /= np.linalg.norm(p)
    new_v[i] = p
  c = child_of_halfedge.reshape(-1, 3)
  i12, i23, i31 = c[:, 0], c[:, 1], c[:, 2]
  i1, i2, i3 = f[:, 0], f[:, 1], f[:, 2]
  new_f = np.stack([
      np.stack([i1, i12, i31], axis=1),
      np.stack([i12, i2, i23], axis=1),
      np.stack([i31, i23, i3], axis=1),
      np.stack([i12, i23, i31], axis=1)], axis=1).reshape(-1, 3)
  return TriangularMesh(np.concatenate([v, new_v], axis=0).astype(np.float32),
                        new_f.astype(np.int32))


def get_hierarchy_of_triangular_meshes_for_sphere(splits: int):
  meshes = [icosahedron()]
  for _ in range(splits):
    meshes.append(_split_faces(meshes[-1]))
  return meshes


def get_last_triangular_mesh_for_sphere(splits: int) -> TriangularMesh:
  return get_hierarchy_of_triangular_meshes_for_sphere(splits)[-1]


def faces_to_edges(faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
  """0->1, 1->2, 2->0 per face, grouped by edge slot (icosahedral_mesh.py:259-281)."""
  assert faces.ndim == 2 and faces.shape[-1] == 3
  senders = np.concatenate([faces[:, 0], faces[:, 1], faces[:, 2]])
  receivers = np.concatenate([faces[:, 1], faces[:, 2], faces[:, 0]])
  return senders, receivers


def max_edge_distance(mesh: TriangularMesh) -> float:
  """gencast/denoiser.py:840-846."""
  s, r = faces_to_edges(mesh.faces)
  return float(np.linalg.norm(mesh.vertices[s] - mesh.vertices[r], axis=-1).max())


# ----------------------------------------------------------------------------
# Spherical helpers (common/model_utils.py:168-203)
# ----------------------------------------------------------------------------

def lat_lon_deg_to_spherical(lat, lon):
  return np.deg2rad(lon), np.deg2rad(90 - lat)


def spherical_to_lat_lon(phi, theta):
  return 90 - np.rad2deg(theta), np.mod(np.rad2deg(phi), 360)


def cartesian_to_spherical(x, y, z):
  with np.errstate(invalid="ignore"):
    return np.arctan2(y, x), np.arccos(z)


def spherical_to_cartesian(phi, theta):
  return (np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta))


def grid_lat_lon_to_coordinates(grid_latitude, grid_longitude) -> np.ndarray:
  """[n_lat],[n_lon] -> [n_lat,n_lon,3] (grid_mesh_connectivity.py:23-38)."""
  phi, theta = np.meshgrid(np.deg2rad(grid_longitude), np.deg2rad(90 - grid_latitude))
  return np.stack([np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta),
                   np.cos(theta)], axis=-1)


def mesh_nodes_lat_lon(mesh: TriangularMesh):
  """float32 lat/lon of mesh vertices (gencast/denoiser.py:419-429)."""
  phi, theta = cartesian_to_spherical(mesh.vertices[:, 0], mesh.vertices[:, 1],
                                      mesh.vertices[:, 2])
  lat, lon = spherical_to_lat_lon(phi=phi, theta=theta)
  return lat.astype(np.float32), lon.astype(np.float32)


def grid_nodes_lat_lon(grid_lat, grid_lon):
  """Flattened node = lat_i * n_lon + lon_j (gencast/denoiser.py:431-441)."""
  lon, lat = np.meshgrid(grid_lon, grid_lat)
  return (lat.reshape([-1]).astype(np.float32), lon.reshape([-1]).astype(np.float32))


# ----------------------------------------------------------------------------
# Connectivity
# ----------------------------------------------------------------------------

def radius_query_indices(*, grid_latitude, grid_longitude, mesh: TriangularMesh,
                         radius: float):
  """Grid->mesh edges with chord distance <= radius.

  Same construction as the reference (cKDTree over mesh vertices, ball query per
  grid point, grid-major edge order; grid_mesh_connectivity.py:40-86).  Inside
  one grid point the mesh indices are sorted ascending so the edge order does not
  depend on the tree's traversal order.
  """
  grid_positions = grid_lat_lon_to_coordinates(grid_latitude, grid_longitude).reshape([-1, 3])
  tree = scipy.spatial.cKDTree(mesh.vertices)
  hits = tree.query_ball_point(x=grid_positions, r=radius)
  g_idx, m_idx = [], []
  for g, nb in enumerate(hits):
    nb = np.sort(np.asarray(nb, dtype=np.int64))
    g_idx.append(np.full(nb.shape, g, dtype=np.int64))
    m_idx.append(nb)
  return np.concatenate(g_idx), np.concatenate(m_idx)


def _closest_point_sqdist_on_triangles(p, a, b, c):
  """Squared distance from points p[N,3] to triangles (a,b,c)[N,3] (Ericson 5.1.5)."""
  ab, ac, ap = b - a, c - a, p - a
  d1 = np.einsum("ij,ij->i", ab, ap)
  d2 = np.einsum("ij,ij->i", ac, ap)
  bp = p - b
  d3 = np.einsum("ij,ij->i", ab, bp)
  d4 = np.einsum("ij,ij->i", ac, bp)
  cp = p - c
  d5 = np.einsum("ij,ij->i", ab, cp)
  d6 = np.einsum("ij,ij->i", ac, cp)
  vc = d1 * d4 - d3 * d2
  vb = d5 * d2 - d1 * d6
  va = d3 * d6 - d5 * d4
  out = np.empty_like(p)
  done = np.zeros(p.shape[0], dtype=bool)

  def put(mask, val):
    m = mask & ~done
    out[m] = val[m]
    done[m] = True

  with np.errstate(divide="ignore", invalid="ignore"):
    put((d1 <= 0) & (d2 <= 0), a)
    put((d3 >= 0) & (d4 <= d3), b)
    put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
    put((d6 >= 0) & (d5 <= d6), c)
    put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
    w_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + w_bc[:, None] * (c - b))
    denom = 1.0 / (va + vb + vc)
    inside = a + (vb * denom)[:, None] * ab + (vc * denom)[:, None] * ac
    put(np.ones_like(done), inside)
  d = p - out
  return np.einsum("ij,ij->i", d, d)


def in_mesh_triangle_indices(*, grid_latitude, grid_longitude, mesh: TriangularMesh,
                             num_candidate_vertices: int = 3):
  """Mesh->grid edges: the 3 vertices of the mesh face closest to each grid point.

  The reference asks trimesh for the Euclidean-closest face
  (grid_mesh_connectivity.py:89-133); trimesh's tie-breaking on shared
  edges/vertices is an implementation accident, so this build
  fixes a documented rule: exact point-triangle distance over every face
  incident to the nearest `num_candidate_vertices` mesh vertices, ties broken
  towards the LOWEST face index (distances compared after rounding to 1e-12).
  Edge order is [3g, 3g+1, 3g+2] = faces[f] vertex order, like the reference.
  """
  pos = grid_lat_lon_to_coordinates(grid_latitude, grid_longitude).reshape([-1, 3])
  pos = pos.astype(np.float64)
  verts = mesh.vertices.astype(np.float64)
  faces = mesh.faces.astype(np.int64)
  n_v = verts.shape[0]
  # vertex -> incident faces (each vertex has 5 or 6)
  flat_v = faces.reshape(-1)
  flat_f = np.repeat(np.arange(faces.shape[0]), 3)
  order = np.argsort(flat_v, kind="stable")
  flat_v, flat_f = flat_v[order], flat_f[order]
  start = np.searchsorted(flat_v, np.arange(n_v + 1))
  max_deg = int((start[1:] - start[:-1]).max())
  inc = np.full((n_v, max_deg), -1, dtype=np.int64)
  for k in range(max_deg):
    has = (start[:-1] + k) < start[1:]
    inc[has, k] = flat_f[start[:-1][has] + k]
  tree = scipy.spatial.cKDTree(verts)
  _, near = tree.query(pos, k=num_candidate_vertices)
  near = near.reshape(pos.shape[0], -1)
  cand = inc[near].reshape(pos.shape[0], -1)             # [G, k*max_deg], -1 = none
  best_d = np.full(pos.shape[0], np.inf)
  best_f = np.full(pos.shape[0], np.iinfo(np.int64).max, dtype=np.int64)
  for j in range(cand.shape[1]):
    fidx = cand[:, j]
    valid = fidx >= 0
    fsafe = np.where(valid, fidx, 0)
    tri = faces[fsafe]
    d = _closest_point_sqdist_on_triangles(pos, verts[tri[:, 0]], verts[tri[:, 1]],
                                           verts[tri[:, 2]])
    d = np.round(np.where(valid, d, np.inf), 12)
    better = (d < best_d) | ((d == best_d) & (fidx < best_f) & valid)
    best_d = np.where(better, d, best_d)
    best_f = np.where(better, fidx, best_f)
  mesh_edge_indices = mesh.faces[best_f].reshape([-1]).astype(np.int64)
  grid_edge_indices = np.repeat(np.arange(pos.shape[0], dtype=np.int64), 3)
  return grid_edge_indices, mesh_edge_indices


# ----------------------------------------------------------------------------
# Structural features
# ----------------------------------------------------------------------------

def node_structural_features(lat: np.ndarray, lon: np.ndarray) -> np.ndarray:
  """(cos theta, cos phi, sin phi), dtype of `lat` (model_utils.py:445-457;
  kwargs from gencast/denoiser.py:241-248: latitude + longitude, no positions)."""
  phi, theta = lat_lon_deg_to_spherical(lat, lon)
  return np.stack([np.cos(theta), np.cos(phi), np.sin(phi)], axis=-1)


def _rotation_to_local(phi: np.ndarray, theta: np.ndarray) -> np.ndarray:
  """R = R_y(pi/2 - theta) . R_z(-phi), float64 [N,3,3].

  Closed form of scipy's extrinsic Rotation.from_euler("zy", [-phi, -theta+pi/2])
  used by the reference (model_utils.py:326-339): puts the reference point at
  longitude 0 then latitude 0.
  """
  az = (-phi).astype(np.float64)
  po = (-theta + np.pi / 2).astype(np.float64)
  ca, sa = np.cos(az), np.sin(az)
  cp, sp = np.cos(po), np.sin(po)
  zeros, ones = np.zeros_like(ca), np.ones_like(ca)
  rz = np.stack([np.stack([ca, -sa, zeros], -1), np.stack([sa, ca, zeros], -1),
                 np.stack([zeros, zeros, ones], -1)], -2)
  ry = np.stack([np.stack([cp, zeros, sp], -1), np.stack([zeros, ones, zeros], -1),
                 np.stack([-sp, zeros, cp], -1)], -2)
  return np.einsum("nij,njk->nik", ry, rz)


def bipartite_edge_structural_features(*, senders_lat, senders_lon, receivers_lat,
                                       receivers_lon, senders, receivers) -> np.ndarray:
  """(|d|, d) / max|d| with d = R_rcv.(p_snd - p_rcv); float64 [E,4].

  model_utils.py:469-495,506-591 with both local-coordinate rotations enabled and
  `edge_normalization_factor=None` (gencast/denoiser.py:469,576).
  """
  s_phi, s_theta = lat_lon_deg_to_spherical(senders_lat, senders_lon)
  r_phi, r_theta = lat_lon_deg_to_spherical(receivers_lat, receivers_lon)
  s_pos = np.stack(spherical_to_cartesian(s_phi, s_theta), axis=-1)
  r_pos = np.stack(spherical_to_cartesian(r_phi, r_theta), axis=-1)
  rot = _rotation_to_local(r_phi, r_theta)[receivers]
  rel = (np.einsum("eji,ei->ej", rot, s_pos[senders])
         - np.einsum("eji,ei->ej", rot, r_pos[receivers]))
  dist = np.linalg.norm(rel, axis=-1, keepdims=True)
  norm = dist.max()
  return np.concatenate([dist / norm, rel / norm], axis=-1)


# ----------------------------------------------------------------------------
# k-hop attention neighbourhoods
# ----------------------------------------------------------------------------

def khop_neighbourhood_csr(num_nodes: int, senders, receivers, k_hop: int):
  """CSR of {(i,j): j within k_hop mesh edges of i} including i itself.

  Equals the sparsity pattern of the reference mask `(A+I)**k`
  (gencast/transformer.py:21-47; gencast/sparse_transformer.py:555) computed as
  boolean reachability, which cannot overflow where the reference's int32 path
  counts can.  Columns are sorted ascending per row.
  """
  a = scipy.sparse.csr_matrix(
      (np.ones(len(senders), dtype=np.int8), (np.asarray(senders), np.asarray(receivers))),
      shape=(num_nodes, num_nodes))
  a = ((a + scipy.sparse.identity(num_nodes, dtype=np.int8, format="csr")) != 0)
  a = a.astype(np.int8).tocsr()
  reach = a.copy()
  for _ in range(k_hop - 1):
    reach = ((reach @ a) != 0).astype(np.int8).tocsr()
  reach.sort_indices()
  return reach.indptr.astype(np.int32), reach.indices.astype(np.int32)


# ----------------------------------------------------------------------------
# Everything the denoiser needs, in one bundle
# ----------------------------------------------------------------------------

@dataclasses.dataclass
class DenoiserGraph:
  """Static graph arrays handed to gc_set_graph (all host numpy)."""
  num_grid_nodes: int
  num_mesh_nodes: int
  g2m_senders: np.ndarray      # [E1] grid index
  g2m_receivers: np.ndarray    # [E1] mesh index
  m2g_senders: np.ndarray      # [E2] mesh index
  m2g_receivers: np.ndarray    # [E2] grid index
  khop_rowptr: np.ndarray      # [M+1]
  khop_cols: np.ndarray        # [nnz]
  grid_struct: np.ndarray      # [G,3] f32
  mesh_struct: np.ndarray      # [M,3] f32
  g2m_edge_struct: np.ndarray  # [E1,4] f32
  m2g_edge_struct: np.ndarray  # [E2,4] f32
  mesh_xyz: np.ndarray         # [M,3] f32
  mesh_senders: np.ndarray     # [Em] 1-hop mesh edges
  mesh_receivers: np.ndarray


def build_denoiser_graph(*, grid_lat, grid_lon, mesh_size: int, attention_k_hop: int,
                         radius_query_fraction_edge_length: float = 0.6) -> DenoiserGraph:
  """What `DenoiserArchitecture._maybe_init` builds (gencast/denoiser.py:343-416,443-600)."""
  mesh = get_last_triangular_mesh_for_sphere(mesh_size)
  grid_lat = np.asarray(grid_lat).astype(np.float32)
  grid_lon = np.asarray(grid_lon).astype(np.float32)
  radius = max_edge_distance(mesh) * radius_query_fraction_edge_length
  g_lat, g_lon = grid_nodes_lat_lon(grid_lat, grid_lon)
  m_lat, m_lon = mesh_nodes_lat_lon(mesh)
  g2m_s, g2m_r = radius_query_indices(grid_latitude=grid_lat, grid_longitude=grid_lon,
                                      mesh=mesh, radius=radius)
  m2g_r, m2g_s = in_mesh_triangle_indices(grid_latitude=grid_lat, grid_longitude=grid_lon,
                                          mesh=mesh)
  e1 = bipartite_edge_structural_features(
      senders_lat=g_lat, senders_lon=g_lon, receivers_lat=m_lat, receivers_lon=m_lon,
      senders=g2m_s, receivers=g2m_r)
  e2 = bipartite_edge_structural_features(
      senders_lat=m_lat, senders_lon=m_lon, receivers_lat=g_lat, receivers_lon=g_lon,
      senders=m2g_s, receivers=m2g_r)
  ms, mr = faces_to_edges(mesh.faces)
  rowptr, cols = khop_neighbourhood_csr(mesh.vertices.shape[0], ms, mr, attention_k_hop)
  return DenoiserGraph(
      num_grid_nodes=int(g_lat.shape[0]), num_mesh_nodes=int(mesh.vertices.shape[0]),
      g2m_senders=g2m_s.astype(np.int32), g2m_receivers=g2m_r.astype(np.int32),
      m2g_senders=m2g_s.astype(np.int32), m2g_receivers=m2g_r.astype(np.int32),
      khop_rowptr=rowptr, khop_cols=cols,
      grid_struct=node_structural_features(g_lat, g_lon).astype(np.float32),
      mesh_struct=node_structural_features(m_lat, m_lon).astype(np.float32),
      g2m_edge_struct=e1.astype(np.float32), m2g_edge_struct=e2.astype(np.float32),
      mesh_xyz=mesh.vertices.astype(np.float32),
      mesh_senders=ms.astype(np.int32), mesh_receivers=mr.astype(np.int32))


def graph_from_reference_arrays(*, grid_lat, grid_lon, mesh_nodes_lat, mesh_nodes_lon, g2m_senders,
                                g2m_receivers, m2g_senders, m2g_receivers, mesh_senders, mesh_receivers,
                                attention_k_hop: int) -> DenoiserGraph:
  """A `DenoiserGraph` from index arrays DUMPED FROM THE REFERENCE (INTEGRATION.md shows the snippet).

  Why: the reference's mesh->grid assignment comes from trimesh's closest-face query
  (common/grid_mesh_connectivity.py:89-133), whose choice for grid points lying exactly on a shared
  mesh edge or vertex is an implementation accident `in_mesh_triangle_indices` cannot promise to
  reproduce (`count_m2g_ties`: 174 of 10 512 points at 2.5 deg / mesh 4).  A checkpoint trained on the
  reference's graph must run on the reference's graph, so its arrays can be passed in as they are:
  mesh node ids are the REFERENCE's (RCM-permuted, gencast/denoiser.py:849-867) -- numbering is
  internal to the model, only the edge sets and the node coordinates they index must be consistent.
  Structural features are recomputed here from the coordinates (common/model_utils.py:364-591).
  """
  grid_lat = np.asarray(grid_lat, np.float32)
  grid_lon = np.asarray(grid_lon, np.float32)
  m_lat = np.asarray(mesh_nodes_lat, np.float32).reshape(-1)
  m_lon = np.asarray(mesh_nodes_lon, np.float32).reshape(-1)
  g_lat, g_lon = grid_nodes_lat_lon(grid_lat, grid_lon)
  G, M = int(g_lat.shape[0]), int(m_lat.shape[0])
  arrs = {}
  for name, a, hi in (("g2m_senders", g2m_senders, G), ("g2m_receivers", g2m_receivers, M),
                      ("m2g_senders", m2g_senders, M), ("m2g_receivers", m2g_receivers, G),
                      ("mesh_senders", mesh_senders, M), ("mesh_receivers", mesh_receivers, M)):
    a = np.asarray(a).reshape(-1)
    if a.size == 0 or a.min() < 0 or a.max() >= hi:
      raise ValueError(f"{name}: indices out of range [0, {hi})")
    arrs[name] = a.astype(np.int32)
  if len(arrs["g2m_senders"]) != len(arrs["g2m_receivers"]) or len(arrs["m2g_senders"]) != len(arrs["m2g_receivers"]) \
      or len(arrs["mesh_senders"]) != len(arrs["mesh_receivers"]):
    raise ValueError("senders / receivers length mismatch")
  e1 = bipartite_edge_structural_features(
      senders_lat=g_lat, senders_lon=g_lon, receivers_lat=m_lat, receivers_lon=m_lon,
      senders=arrs["g2m_senders"], receivers=arrs["g2m_receivers"])
  e2 = bipartite_edge_structural_features(
      senders_lat=m_lat, senders_lon=m_lon, receivers_lat=g_lat, receivers_lon=g_lon,
      senders=arrs["m2g_senders"], receivers=arrs["m2g_receivers"])
  rowptr, cols = khop_neighbourhood_csr(M, arrs["mesh_senders"], arrs["mesh_receivers"], attention_k_hop)
  phi, theta = lat_lon_deg_to_spherical(m_lat, m_lon)
  xyz = np.stack(spherical_to_cartesian(phi, theta), axis=-1)
  return DenoiserGraph(
      num_grid_nodes=G, num_mesh_nodes=M, g2m_senders=arrs["g2m_senders"], g2m_receivers=arrs["g2m_receivers"],
      m2g_senders=arrs["m2g_senders"], m2g_receivers=arrs["m2g_receivers"], khop_rowptr=rowptr, khop_cols=cols,
      grid_struct=node_structural_features(g_lat, g_lon).astype(np.float32),
      mesh_struct=node_structural_features(m_lat, m_lon).astype(np.float32),
      g2m_edge_struct=e1.astype(np.float32), m2g_edge_struct=e2.astype(np.float32),
      mesh_xyz=xyz.astype(np.float32), mesh_senders=arrs["mesh_senders"], mesh_receivers=arrs["mesh_receivers"])


def load_reference_graph(path: str, *, grid_lat, grid_lon, attention_k_hop: int) -> DenoiserGraph:
  """`graph_from_reference_arrays` from the .npz written by the INTEGRATION.md dump snippet."""
  with np.load(path) as z:
    return graph_from_reference_arrays(
        grid_lat=grid_lat, grid_lon=grid_lon, attention_k_hop=attention_k_hop,
        **{k: z[k] for k in ("mesh_nodes_lat", "mesh_nodes_lon", "g2m_senders", "g2m_receivers", "m2g_senders",
                             "m2g_receivers", "mesh_senders", "mesh_receivers")})


def count_m2g_ties(*, grid_latitude, grid_longitude, mesh: TriangularMesh) -> int:
  """Grid points whose closest mesh triangle is NOT unique (they sit on a shared edge or vertex, the
  poles for example): for those the reference's trimesh query and `in_mesh_triangle_indices` (lowest
  face index) may pick different faces, i.e. different mesh2grid senders.  Same candidate set and the
  same 1e-12 distance rounding as `in_mesh_triangle_indices`."""
  pos = grid_lat_lon_to_coordinates(grid_latitude, grid_longitude).reshape([-1, 3]).astype(np.float64)
  verts = mesh.vertices.astype(np.float64)
  faces = mesh.faces.astype(np.int64)
  tree = scipy.spatial.cKDTree(verts)
  _, near = tree.query(pos, k=3)
  incident = [[] for _ in range(verts.shape[0])]
  for f, tri in enumerate(faces):
    for v in tri:
      incident[v].append(f)
  ties = 0
  for s0 in range(0, pos.shape[0], 4096):
    p = pos[s0:s0 + 4096]
    cands = [sorted({f for v in row for f in incident[v]}) for row in near[s0:s0 + 4096]]
    width = max(len(c) for c in cands)
    cand = np.array([c + [-1] * (width - len(c)) for c in cands])
    d_all = np.full(cand.shape, np.inf)
    for j in range(width):
      valid = cand[:, j] >= 0
      tri = faces[np.where(valid, cand[:, j], 0)]
      d = _closest_point_sqdist_on_triangles(p, verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]])
      d_all[:, j] = np.round(np.where(valid, d, np.inf), 12)
    ties += int(((d_all == d_all.min(axis=1, keepdims=True)).sum(axis=1) > 1).sum())
  return ties
