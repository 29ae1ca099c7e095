classdef MonteCarloTreeSearchHip < OptimizerInterface
    % MONTECARLOTREESEARCHHIP  The sampled optimizer on an AMD MI355X through libpdmpc_hip.so.
    %
    % Drop-in replacement for MonteCarloTreeSearch (hlc/optimizer/graph_search/MonteCarloTreeSearch.m): same
    % run_optimizer signature, same random stream (mt19937ar seeded with time_step + vehicle_index, :32, generated
    % inside the library), same ControlResultsInfo fields.  Selected with options.optimizer_type =
    % OptimizerType.HipSampled (one more enum member and one more case in OptimizerInterface.get_optimizer,
    % see INTEGRATION.md).
    %
    % Shipped as source like GraphSearchHip.m; the tested twin is pdmpc.optimizer.MonteCarloTreeSearchHip.

    properties (Access = private)
        handle uint64 = uint64(0);
        mpa_uploaded (1, 1) logical = false;
    end

    methods

        function obj = MonteCarloTreeSearchHip(options)
            obj = obj@OptimizerInterface();
            checker = double(options.are_any_obstacles_non_convex); % OptimizerInterface.m:36-46
            obj.handle = pdmpc_mex('create', options.Hp, checker, options.dt_seconds);
        end

        function delete(obj)

            if obj.handle ~= 0
                pdmpc_mex('destroy', obj.handle);
            end

        end

        function info = run_optimizer(obj, vehicle_index, iter, mpa, options, time_step)
            assert(iter.amount == 1); % MonteCarloTreeSearch.m:50

            if ~obj.mpa_uploaded
                pdmpc_mex('upload_mpa', obj.handle, mpa.transition_matrix_single, mpa.maneuvers);
                obj.mpa_uploaded = true;
            end

            Hp = options.Hp;
            info = ControlResultsInfo(iter.amount, Hp);
            out = pdmpc_mex('plan_sampled', obj.handle, pdmpc_iter_struct(iter), time_step + vehicle_index);

            info.n_expanded = out.n_expanded; % MonteCarloTreeSearch.m:209
            info.is_exhausted = out.status ~= 0; % :212-215

            if info.is_exhausted
                return
            end

            % :217-248: the tree of the reference holds only the chosen descent's poses; rebuild that view
            tree = Tree();
            tree.x = out.path_nodes(:, 1)';
            tree.y = out.path_nodes(:, 2)';
            tree.yaw = out.path_nodes(:, 3)';
            tree.trim = out.path_nodes(:, 4)';
            tree.g = out.path_nodes(:, 5)';
            tree.h = out.path_nodes(:, 6)';
            tree.k = out.path_nodes(:, 7)';
            tree.parent = uint32(0:Hp);
            info.tree = tree;
            info.tree_path = 1:(Hp + 1);
            info.y_predicted = out.y_predicted(1:Hp, :)';
            info.shapes = arrayfun(@(k) squeeze(out.shapes(k, :, 1:out.shape_cols(k))), 1:Hp, UniformOutput = false);
            info.predicted_trims = out.predicted_trims(1:Hp);
            info.needs_fallback = false;
        end

    end

end
