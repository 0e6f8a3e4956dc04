// Append to src/raytracing/model.rs (Triangle's fields, Model.obj and the ModelGrid arrays of model.rs:5,9-14,205-210 are
// private to that module).
impl Triangle {
    /// The three corners, as Model::get_triangle (model.rs:108-122) produced them.
    pub fn vertices(&self) -> (Vec3, Vec3, Vec3) {
        (self.0, self.1, self.2)
    }
}

impl Model {
    /// Number of triangles: three u16 indices each (model.rs:141).
    pub fn n_triangles(&self) -> usize {
        self.obj.indices.len() / 3
    }
}

impl ModelGrid {
    /// ModelGrid.offset_array: inclusive prefix sums of the cell list lengths (model.rs:185-193).
    pub fn offset_array(&self) -> &[usize] {
        &self.offset_array
    }
    /// ModelGrid.triangle_indices (model.rs:191).
    pub fn triangle_indices(&self) -> &[usize] {
        &self.triangle_indices
    }
}
