(* GPHIP.wl -- thin Wolfram-Language host package for the MI355X GP path.

   Load AFTER the reference package (BayesianInference/Kernel/BayesianInference.wl:11-19): it reuses the
   reference's own inferenceObject, $MachineLogZero, dataNormalForm and defineInferenceProblem and swaps
   (1) the "LogLikelihoodFunction" closure (seam at BayesianGaussianProcess.wl:249, 293-294) and
   (2) the prediction down-value for HIP-backed objects (BayesianGaussianProcess.wl:343-376),
   keeping every key of "GaussianProcessData"/"ModelFunctions" in the SHAPE the reference defines
   (BayesianGaussianProcess.wl:257-262, 308, 314-321) so that reference code reading the object --
   predictFromGaussianProcess' own loop (:358-368), regressionPlot1D -- keeps working on it:
       "KernelFunction", "NuggetFunction", "MeanFunction"   theta |-> pure function   (expressionToFunction, :257-262)
       "CovarianceFunction"                                 theta |-> N x N matrix    (compiledCovarianceMatrix, :265-270)
       "InverseCovarianceFunction"                          theta |-> <|"Inverse" -> solver, "LogDet" -> real|>  (:137-141, 308)
   All numerics are in libgphip (include/gphip.h) behind the LibraryLink shim (csrc/librarylink_shim.cpp).
   This file cannot be executed in the build containers (no Wolfram kernel).  What IS tested there:
   the shim is compiled against a stub WolframLibrary.h and every gphip_wl_* entry point is driven through a
   fake WolframLibraryData on the GPU (tests/test_gpu_wl_shim.py); tests/test_wl_package.py checks that every
   LibraryFunctionLoad below names an exported shim function with the same argument count; and every call made
   here is mirrored 1:1 by bayesianinference_amd/gaussian_process.py, which the parity tests exercise. *)

BeginPackage["GPHIP`", {"BayesianUtilities`", "BayesianStatistics`", "BayesianGaussianProcess`"}]

defineGaussianProcessHIP::usage = "defineGaussianProcessHIP[X -> Y, kernel, nugget, meanFunction, variables, prior, opts] has the argument list of defineGaussianProcess (BayesianGaussianProcess.wl:228-234) and builds the same inferenceObject with the log-likelihood evaluated on the GPU. kernel: a named kernel \"SE\", \"SEARD\", \"Matern52\", \"Matern52ARD\", \"Matern32\", \"Matern32ARD\", \"RQ\", \"RQARD\", a composed form \"term + term\", \"term * term\", optionally followed by \" + Const\" (e.g. \"SE + Const\"), or None (null kernel); ANY OTHER kernel expression falls through to the reference's own defineGaussianProcess. nugget: \"Constant\" (Function[sn^2]) or any expression / function of the point in the parameter symbols (evaluated on the host per theta, the values go to the GPU). meanFunction: None, \"Constant\" or any expression / function of the point. variables in the library's order: {term 1: l.., (alpha), sf}, {term 2 ..}, {c}, {sn}, {mu}. The short form defineGaussianProcessHIP[X -> Y, kernelName, variables, prior, opts] takes the constant nugget and \"ConstantMean\" -> False | True. Options: \"Precision\" -> \"Double\" | \"Single\", \"Devices\" -> Automatic | {0, 1, ..}, \"LibraryOptions\" -> {\"panel\" -> 4, ..}.";
nestedSamplingHIP::usage = "nestedSamplingHIP[obj, opts] runs the native batched nested-sampling driver of the library (lock-step walkers: one batched likelihood call per Metropolis step) on a HIP-backed GP object whose prior is a product of UniformDistribution's (or \"PriorKinds\" -> {0 | 1 ..}, 1 = log-uniform over the parameter's range) or ANY product of univariate distributions (its factors' log densities travel as tables, the starting pool is drawn from the prior here) and returns the object joined with the result, in the shape nestedSampling returns (the reference's own evidenceSampling post-processes the samples). Takes the options of nestedSampling plus \"Walkers\" -> 32 and \"Seed\" -> 0. Joint (non-separable) priors fall through to nestedSampling[obj], which drives the same GPU closure one theta at a time.";
defineGaussianProcessHIP::nonnative = "Kernel `1` is not one of the named or composed kernels of the library: give the nugget and the mean function as expressions in the parameter symbols (not \"Constant\"); the kernel is then compiled for the device from its CForm, or runs on the reference's own path if it cannot be printed as C.";
hipKernelFunction::usage = "hipKernelFunction[kernelName, d] gives theta |-> Function[{p, q}, ..], the exact WL form of the named kernel (what one would hand to the reference's defineGaussianProcess for the same model).";
$GPHIPLibrary::usage = "Path of the LibraryLink shim (libgphip_wl).";

Begin["`Private`"]

$GPHIPLibrary = FindLibrary["libgphip_wl"];
kernelIds = <|"SE" -> 0, "SEARD" -> 1, "Matern52" -> 2, "Matern52ARD" -> 3, None -> 4,
	"Matern32" -> 5, "Matern32ARD" -> 6, "RQ" -> 7, "RQARD" -> 8|>;
nLengthScales[name_, d_] := Switch[name, "SE" | "Matern52" | "Matern32" | "RQ", 1, "SEARD" | "Matern52ARD" | "Matern32ARD" | "RQARD", d, _, 0];
termParams[name_, d_] := nLengthScales[name, d] + Boole[StringStartsQ[name, "RQ"]] + 1;     (* l.., (alpha), sf *)

(* kernel spec -> {term1, op (0 none, 1 sum, 2 product), term2 | None, offset (0 | 1)} or $Failed (not native).
   Grammar (spaces ignored): term, "term + term" or "term * term", optionally followed by "+ Const"
   -- include/gphip.h GPHIP_KERNEL_COMPOSE *)
parseKernel[None] := {None, 0, None, 0};
parseKernel[name_String] := Module[{key = StringDelete[name, Whitespace], offset = 0, parts, op = 0},
	If[ StringEndsQ[key, "+Const", IgnoreCase -> True], key = StringDrop[key, -6]; offset = 1];
	parts = Which[
		StringContainsQ[key, "+"], op = 1; StringSplit[key, "+", 2],
		StringContainsQ[key, "*"], op = 2; StringSplit[key, "*", 2],
		True, {key}
	];
	If[ AllTrue[parts, KeyExistsQ[kernelIds, #] && # =!= None &] && Length[parts] === If[op === 0, 1, 2],
		{parts[[1]], op, If[op === 0, None, parts[[2]]], offset},
		$Failed
	]
];
parseKernel[_] := $Failed;
kernelCode[{None, __}] := 4;
kernelCode[{t1_, 0, _, 0}] := kernelIds[t1];
kernelCode[{t1_, op_, t2_, offset_}] := BitOr[kernelIds[t1], BitShiftLeft[If[t2 === None, 0, kernelIds[t2]], 8],
	BitShiftLeft[op, 16], BitShiftLeft[offset, 20], BitShiftLeft[1, 24]];
(* position of sigma_n in theta (1-based) for a parsed spec *)
nuggetIndex[{None, __}, d_] := 1;
nuggetIndex[{t1_, op_, t2_, offset_}, d_] := termParams[t1, d] + If[op === 0, 0, termParams[t2, d]] + offset + 1;

(* ---- LibraryLink bindings (argument lists are checked against the shim by tests/test_wl_package.py) ---- *)
gpCreate   := gpCreate   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_create",
	{{Real, 2, "Constant"}, {Real, 1, "Constant"}, Integer, Integer, Integer, {Integer, 1, "Constant"}}, Integer];
gpCreateCustom := gpCreateCustom = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_create_custom",
	{{Real, 2, "Constant"}, {Real, 1, "Constant"}, "UTF8String", Integer, Integer, Integer, Integer}, Integer];
gpSetOpt   := gpSetOpt   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_set_option", {Integer, "UTF8String", Real}, Integer];
gpLogLik   := gpLogLik   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik",
	{Integer, {Real, 1, "Constant"}}, {Real, 1}];        (* {value, info} *)
gpLogLikB  := gpLogLikB  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik_batch",
	{Integer, {Real, 2, "Constant"}}, {Real, 2}];        (* {{value, info}..} *)
gpGrad     := gpGrad     = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik_grad",
	{Integer, {Real, 1, "Constant"}}, {Real, 1}];        (* {value, info, grad..} *)
gpFit      := gpFit      = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_fit",
	{Integer, {Real, 1, "Constant"}}, Integer];          (* info *)
gpSolve    := gpSolve    = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_solve",
	{Integer, {Real, _, "Constant"}}, {Real, _}];        (* vector (N) or matrix (N x m), same shape back *)
gpLogDet   := gpLogDet   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_logdet", {Integer}, Real];
gpPredict  := gpPredict  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict",
	{Integer, {Real, 2, "Constant"}}, {Real, 2}];        (* {means, variances} *)
gpPredictS := gpPredictS = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict_samples",
	{Integer, {Real, 2, "Constant"}, {Real, 2, "Constant"}}, {Real, 3}];   (* {means, variances}, each S x M *)
gpCov      := gpCov      = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_covariance",
	{Integer, {Real, 1, "Constant"}}, {Real, 2}];
gpCross    := gpCross    = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_cross_covariance",
	{Integer, {Real, 1, "Constant"}, {Real, 2, "Constant"}}, {Real, 2}];   (* (N+1) x M: k on top, kappa last row *)
gpDestroy  := gpDestroy  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_destroy", {Integer}, Integer];
gpDevices  := gpDevices  = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_device_count", {}, Integer];
(* point-dependent nugget[x] / meanFunction[x]: VALUES per theta and point; {} = the constant form *)
gpLogLikBPW := gpLogLikBPW = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_loglik_batch_pw",
	{Integer, {Real, 2, "Constant"}, {Real, _, "Constant"}, {Real, _, "Constant"}}, {Real, 2}];
gpFitPW    := gpFitPW    = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_fit_pw",
	{Integer, {Real, 1, "Constant"}, {Real, _, "Constant"}, {Real, _, "Constant"}}, Integer];
gpPredictSPW := gpPredictSPW = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_predict_samples_pw",
	{Integer, {Real, 2, "Constant"}, {Real, _, "Constant"}, {Real, _, "Constant"}, {Real, 2, "Constant"},
	 {Real, _, "Constant"}, {Real, _, "Constant"}}, {Real, 3}];
gpNested   := gpNested   = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_nested_sampling",
	{Integer, {Real, 2, "Constant"}, {Integer, 1, "Constant"}, {Real, 1, "Constant"}, {Real, _, "Constant"}}, {Real, 2}];
gpNestedTab := gpNestedTab = LibraryFunctionLoad[$GPHIPLibrary, "gphip_wl_nested_sampling_tab",
	{Integer, {Real, 2, "Constant"}, {Real, 2, "Constant"}, {Real, 1, "Constant"}, {Real, 2, "Constant"}}, {Real, 2}];

(* value -> machine real; info != 0 or a LibraryFunctionError -> $MachineLogZero, exactly what
   Catch[..., "MatInv"] yields in the reference closure (BayesianGaussianProcess.wl:298-304). *)
toLogLik[{val_Real, info_Real}] := If[info == 0., Clip[val, {-Abs[$MachineLogZero], Abs[$MachineLogZero]}], $MachineLogZero];
toLogLik[_] := $MachineLogZero;

(* ---- the named kernels as the reference would see them: theta |-> pure function ---- *)
hipKernelFunction["SE", d_] := Function[theta,
	With[{l = theta[[1]], sf = theta[[2]]},
		Function[{p, q}, sf^2 Exp[-Total[(p - q)^2]/(2 l^2)]]]];
hipKernelFunction["SEARD", d_] := Function[theta,
	With[{ls = theta[[;; d]], sf = theta[[d + 1]]},
		Function[{p, q}, sf^2 Exp[-Total[((p - q)/ls)^2]/2]]]];
hipKernelFunction["Matern52", d_] := Function[theta,
	With[{l = theta[[1]], sf = theta[[2]]},
		Function[{p, q}, With[{s = Sqrt[Total[(p - q)^2]]/l}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]]];
hipKernelFunction["Matern52ARD", d_] := Function[theta,
	With[{ls = theta[[;; d]], sf = theta[[d + 1]]},
		Function[{p, q}, With[{s = Sqrt[Total[((p - q)/ls)^2]]}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]]];
hipKernelFunction["Matern32", d_] := Function[theta,
	With[{l = theta[[1]], sf = theta[[2]]},
		Function[{p, q}, With[{s = Sqrt[Total[(p - q)^2]]/l}, sf^2 (1 + Sqrt[3] s) Exp[-Sqrt[3] s]]]]];
hipKernelFunction["Matern32ARD", d_] := Function[theta,
	With[{ls = theta[[;; d]], sf = theta[[d + 1]]},
		Function[{p, q}, With[{s = Sqrt[Total[((p - q)/ls)^2]]}, sf^2 (1 + Sqrt[3] s) Exp[-Sqrt[3] s]]]]];
hipKernelFunction["RQ", d_] := Function[theta,
	With[{l = theta[[1]], a = theta[[2]], sf = theta[[3]]},
		Function[{p, q}, sf^2 (1 + Total[(p - q)^2]/(2 a l^2))^(-a)]]];
hipKernelFunction["RQARD", d_] := Function[theta,
	With[{ls = theta[[;; d]], a = theta[[d + 1]], sf = theta[[d + 2]]},
		Function[{p, q}, sf^2 (1 + Total[((p - q)/ls)^2]/(2 a))^(-a)]]];
hipKernelFunction[None, d_] := Function[theta, Function[0]];      (* nullKernelPattern, BayesianGaussianProcess.wl:25 *)
(* composed forms: each term reads its own slice of theta; c sits behind the terms *)
hipKernelFunction[{t1_, 0, _, 0}, d_] := hipKernelFunction[t1, d];
hipKernelFunction[{t1_, op_, t2_, offset_}, d_] := With[{n1 = termParams[t1, d], n2 = If[op === 0, 0, termParams[t2, d]]},
	Function[theta,
		With[{
			k1 = hipKernelFunction[t1, d][theta[[;; n1]]],
			k2 = If[op === 0, None, hipKernelFunction[t2, d][theta[[n1 + 1 ;; n1 + n2]]]],
			c = If[offset === 1, theta[[n1 + n2 + 1]], 0]
		},
			Switch[op,
				0, Function[{p, q}, c + k1[p, q]],
				1, Function[{p, q}, c + k1[p, q] + k2[p, q]],
				2, Function[{p, q}, c + k1[p, q] k2[p, q]]
			]
		]
	]
];
hipNuggetFunction[spec_, d_] := With[{i = nuggetIndex[spec, d]},
	Function[theta, With[{sn = theta[[i]]}, Function[sn^2]]]];
hipMeanFunction[spec_, d_, False] := Function[theta, Function[0]];
hipMeanFunction[spec_, d_, True] := With[{i = nuggetIndex[spec, d] + 1},
	Function[theta, With[{mu = theta[[i]]}, Function[mu]]]];

(* the handle keeps ONE factor resident: refit only when theta changed since the last fit.  Every other call into the
   library reuses the handle's workspace, so it first forgets the fit (touch).  fit = gpFit[h, #]& or the point-dependent
   form that also hands over nugget / mean values. *)
$fitted = <||>;
touch[h_] := ($fitted[h] = None);
ensureFit[h_, theta_, fit_] := If[ Lookup[$fitted, h, None] === theta,
	0,
	With[{info = fit[theta]},
		$fitted[h] = If[info === 0, theta, None];
		info
	]
];

(* ---- an arbitrary kernel as C text.  The kernel expression (in the parameter symbols) is applied to two symbolic points;
   the result must be free of anything CForm cannot print as elementary arithmetic.  Coordinates become X(k) / Y(k), the
   parameter symbols P(k) in the order of `variables` (k from 0): the argument convention of gphip_create_custom.  The
   library's theta for such a handle is {p.., sn}: the reference's nugget arrives as values per point, so the sn slot is a
   dummy 1. appended by customLift. ---- *)
customKernelSpec[kerf_, vars_List, d_Integer] := Module[{xs, ys, ps, expr, str},
	(* plain symbols gphipXc0, gphipXc1, .. stand for the coordinates (N[] would turn an index argument into a real) *)
	xs = Table[Symbol["GPHIP`Private`gphipXc" <> ToString[k]], {k, 0, d - 1}];
	ys = Table[Symbol["GPHIP`Private`gphipYc" <> ToString[k]], {k, 0, d - 1}];
	ps = Table[Symbol["GPHIP`Private`gphipPc" <> ToString[k]], {k, 0, Length[vars] - 1}];
	expr = Quiet @ Check[kerf[xs, ys], $Failed];
	If[ expr === $Failed || !FreeQ[expr, _Function | _Slot | _Piecewise | _If | _Which | _List | _Dot], Return[$Failed]];
	expr = N[expr /. Thread[vars -> ps]];
	(* anything left that is neither a coordinate / parameter stand-in nor a System` function cannot be printed as C *)
	If[ Cases[expr, sym_Symbol /; !MemberQ[Join[xs, ys, ps], sym] && Context[sym] =!= "System`", {0, Infinity}, Heads -> True] =!= {},
		Return[$Failed]];
	(* (a symbol of a context that is not on $ContextPath prints with its context: GPHIP_Private_gphipXc0) *)
	str = StringReplace[ToString[CForm[expr]], {
		RegularExpression["[A-Za-z0-9_`]*gphipXc(\\d+)"] -> "X($1)", RegularExpression["[A-Za-z0-9_`]*gphipYc(\\d+)"] -> "Y($1)",
		RegularExpression["[A-Za-z0-9_`]*gphipPc(\\d+)"] -> "P($1)"}];
	{"Custom", "return " <> str <> ";", kerf, Length[vars], vars}
];
customSpecQ[spec_] := MatchQ[spec, {"Custom", _String, _, _Integer, _List}];
customLift[spec_][theta_] := If[ customSpecQ[spec],
	If[MatrixQ[theta], ArrayFlatten[{{theta, ConstantArray[1., {Length[theta], 1}]}}], Join[theta, {1.}]],
	theta
];
hipKernelFunction[{"Custom", _, kerf_, _, vars_}, d_] := Function[theta, kerf /. Thread[vars -> theta]];
hipNuggetFunction[{"Custom", __}, d_] := Function[theta, Function[1.]];       (* the dummy sn^2 the library puts on the diagonal *)

Options[defineGaussianProcessHIP] = {"ConstantMean" -> False, "Precision" -> "Double", "Devices" -> Automatic, "LibraryOptions" -> {}};

constantQ[f_] := MatchQ[f, "Constant" | Automatic];
zeroMeanQ[f_] := MatchQ[f, None | 0 | 0. | Function[0] | (0 &)];

(* The reference's own argument list (BayesianGaussianProcess.wl:228-234).  A kernel that is not one of the named /
   composed forms falls through to the reference's defineGaussianProcess with the caller's arguments untouched: the
   object is then the reference's own (interpreted kernel build + LinearSolve), exactly what SURVEY.md section 7 promises. *)
defineGaussianProcessHIP[
	dataIn_List?(MatrixQ[#, NumericQ]&) -> dataOut_List?(MatrixQ[#, NumericQ]&),
	kerf_, nugf_, meanf_,
	variables : {{_Symbol, _, _}..},
	variablePrior_,
	rest___Rule
] /; Dimensions[dataOut][[2]] === 1 && Length[dataIn] === Length[dataOut] := Module[{
	spec = parseKernel[kerf],
	own = "ConstantMean" | "Precision" | "Devices" | "LibraryOptions"
},
	If[ spec === $Failed,
		(* not a named kernel: the reference's meaning of every argument (so the nugget and the mean must be given the
		   reference's way too, not through the "Constant" shorthands of the named kernels) *)
		If[ StringQ[nugf] || nugf === Automatic || StringQ[meanf],
			Message[defineGaussianProcessHIP::nonnative, kerf];
			Return[inferenceObject[$Failed]]
		];
		(* ANY pure function of two points in the parameter symbols (BGP:29-33): printed with CForm and compiled by the
		   library at run time into its own kernel build (gphip_create_custom).  Only when that is not possible -- the
		   expression does not reduce to elementary functions of the coordinates, or the text does not compile -- does the
		   object fall through to the reference's interpreted path *)
		With[{custom = customKernelSpec[kerf, variables[[All, 1]], Dimensions[dataIn][[2]]]},
			If[ custom =!= $Failed,
				With[{obj = Quiet @ Check[hipGaussianProcess[dataIn -> dataOut, custom, nugf, meanf, variables, variablePrior, rest], $Failed]},
					If[ obj =!= $Failed && !MatchQ[obj, inferenceObject[$Failed]], Return[obj]]
				]
			]
		];
		Return @ defineGaussianProcess[dataIn -> dataOut, kerf, nugf, meanf, variables, variablePrior,
			Sequence @@ FilterRules[{rest}, Except[own]]]
	];
	hipGaussianProcess[dataIn -> dataOut, spec, nugf, meanf, variables, variablePrior, rest]
];
(* short form: named kernel, constant nugget, "ConstantMean" option *)
defineGaussianProcessHIP[
	data : (_List -> _List),
	kernelName : (_String | None),
	variables : {{_Symbol, _, _}..},
	variablePrior_,
	rest___Rule
] := defineGaussianProcessHIP[data, kernelName, "Constant", If[TrueQ[Lookup[{rest}, "ConstantMean", False]], "Constant", None],
	variables, variablePrior, rest];
defineGaussianProcessHIP[___] := inferenceObject[$Failed];

hipGaussianProcess[dataIn_ -> dataOut_, spec_, nugf_, meanf_, variables_, variablePrior_, rest___Rule] := Module[{
	h, loglik, invCov, fit, values,
	d = Dimensions[dataIn][[2]],
	vars = variables[[All, 1]],
	inputData = Developer`ToPackedArray[N @ dataIn],
	constMean = constantQ[meanf],
	nugPW = !constantQ[nugf],                               (* nugget[points[[i]]] evaluated on the host (BGP:37) *)
	meanPW = !constantQ[meanf] && !zeroMeanQ[meanf],         (* meanFunction /@ inputData (BGP:300) *)
	nugget, mean, pw,
	dtype = If[Lookup[{rest}, "Precision", "Double"] === "Single", 32, 64],
	(* sub-kernels of parallelNestedSampling pick their own GPU (BayesianStatistics.wl:1349) among the devices the
	   machine really has; a list of several ordinals makes ONE multi-device handle: the library shards a large
	   factorisation over them *)
	devices = Replace[Lookup[{rest}, "Devices", Automatic], {Automatic :> {Mod[$KernelID, Max[gpDevices[], 1]]}, i_Integer :> {i}}],
	own = "ConstantMean" | "Precision" | "Devices" | "LibraryOptions"
},
	h = If[ customSpecQ[spec],
		gpCreateCustom[inputData, N @ Flatten[dataOut], spec[[2]], spec[[4]], 0, dtype, First[devices]],
		gpCreate[inputData, N @ Flatten[dataOut], kernelCode[spec], Boole[constMean], dtype, devices]
	];
	If[ !IntegerQ[h] || h < 0, Return[inferenceObject[$Failed]]];
	KeyValueMap[gpSetOpt[h, #1, N[#2]]&, Association @ Lookup[{rest}, "LibraryOptions", {}]];
	(* theta |-> function of the point, exactly as the reference builds them (expressionToFunction, BGP:257-262) *)
	nugget = If[nugPW, expressionToFunction[nugf, vars -> paramVector], hipNuggetFunction[spec, d]];
	mean = Which[meanPW, expressionToFunction[meanf, vars -> paramVector], True, hipMeanFunction[spec, d, constMean]];
	pw = nugPW || meanPW;
	(* values of a point function for every theta of a batch: B x Length[pts], or {} for the constant form *)
	values[f_, on_, thetas_, pts_] := If[on, Developer`ToPackedArray @ N @ Map[Function[th, f[th] /@ pts], thetas], {}];
	loglik = Function[theta,
		touch[h];
		Which[
			pw, With[{ths = If[MatrixQ[theta], N @ theta, {N @ theta}]},
				With[{res = toLogLik /@ gpLogLikBPW[h, customLift[spec][ths], values[mean, meanPW, ths, inputData], values[nugget, nugPW, ths, inputData]]},
					If[MatrixQ[theta], res, First[res]]]],
			MatrixQ[theta], toLogLik /@ gpLogLikB[h, N @ theta],
			True, toLogLik @ gpLogLik[h, N @ theta]
		]
	];
	fit = If[ pw,
		Function[th, gpFitPW[h, customLift[spec][th], Flatten @ values[mean, meanPW, {th}, inputData], Flatten @ values[nugget, nugPW, {th}, inputData]]],
		Function[th, gpFit[h, th]]
	];
	(* matrixInverseAndDet[covarianceFunction[theta]] (BayesianGaussianProcess.wl:130-141, 308): an Association with
	   a solver that takes a vector or a matrix (:194, :410, :416) and the log-determinant; singular K Throws the
	   sentinel with tag "MatInv" exactly like :133 *)
	invCov = Function[theta,
		With[{th = N @ theta},
			If[ ensureFit[h, th, fit] =!= 0, Throw[$MachineLogZero, "MatInv"]];
			<|
				"Inverse" -> Function[b, If[ensureFit[h, th, fit] =!= 0, Throw[$MachineLogZero, "MatInv"]]; gpSolve[h, N @ b]],
				"LogDet" -> gpLogDet[h]
			|>
		]
	];
	defineInferenceProblem[                                (* same keys as BayesianGaussianProcess.wl:310-325 *)
		"Data" -> dataNormalForm[dataIn -> dataOut],
		"PriorDistribution" -> variablePrior,
		"Parameters" -> variables,
		"GaussianProcessData" -> <|
			"ModelFunctions" -> <|
				"KernelFunction" -> hipKernelFunction[spec, d],
				"NuggetFunction" -> nugget,
				"MeanFunction" -> mean,
				(* the library builds K with the constant sn^2 on the diagonal; a point-dependent nugget replaces it here *)
				"CovarianceFunction" -> Function[theta, touch[h];
					With[{K = gpCov[h, customLift[spec][N @ theta]]},
						If[nugPW, K + DiagonalMatrix[(nugget[theta] /@ inputData) - hipNuggetFunction[spec, d][theta][]], K]]],
				"InverseCovarianceFunction" -> invCov
			|>,
			"KernelName" -> spec,
			"PointwiseFunctions" -> {meanPW, nugPW},
			"ThetaLift" -> customLift[spec],                       (* identity for the named kernels *)
			"HIPHandle" -> h
		|>,
		Sequence @@ FilterRules[{rest}, Except[own]],
		"LogLikelihoodGradientFunction" -> If[pw, Missing["PointDependentFunctions"], Function[theta, touch[h]; gpGrad[h, N @ theta]]],
		"LogLikelihoodFunction" -> loglik
	]
];

(* ---- prediction for HIP-backed objects: same return shape as BayesianGaussianProcess.wl:343-376 ---- *)
hipObjectQ = Function[AssociationQ[#] && KeyExistsQ[#, "Samples"] &&
	KeyExistsQ[Lookup[#, "GaussianProcessData", <||>], "HIPHandle"]];

hipPredict[result_, pts_List] := Module[{
	h = result["GaussianProcessData", "HIPHandle"],
	points = dataNormalForm[pts],
	train = result["Data"][[1]],
	weights = Values @ result[["Samples", All, "CrudePosteriorWeight"]],
	thetas = N @ Values @ result[["Samples", All, "Point"]],
	mf = result["GaussianProcessData", "ModelFunctions"],
	pwFlags = Lookup[result["GaussianProcessData"], "PointwiseFunctions", {False, False}],
	vals, perSample
},
	If[ Dimensions[points][[2]] =!= Dimensions[train][[2]], Return[$Failed]];       (* test points of the wrong width *)
	vals[f_, on_, at_] := If[on, Developer`ToPackedArray @ N @ Map[Function[th, f[th] /@ at], thetas], {}];
	(* one batched call: every posterior sample is factored and solved in its own workspace slot; a sample whose K
	   is singular comes back as NaN rows.  Point-dependent nugget / mean functions are evaluated per sample at the
	   training AND the test points (BGP:113, 408) *)
	perSample = With[{mv = If[ Or @@ pwFlags,
			gpPredictSPW[h, Lookup[result["GaussianProcessData"], "ThetaLift", Identity][thetas], vals[mf["MeanFunction"], pwFlags[[1]], train], vals[mf["NuggetFunction"], pwFlags[[2]], train],
				N @ points, vals[mf["MeanFunction"], pwFlags[[1]], points], vals[mf["NuggetFunction"], pwFlags[[2]], points]],
			gpPredictS[h, thetas, N @ points]
		]},
		MapThread[
			Function[{mus, vars}, MapThread[NormalDistribution, {mus, Sqrt[vars]}]],
			{mv[[1]], mv[[2]]}
		]
	];
	touch[h];                                                (* the batched pass reused the handle's workspace *)
	AssociationThread[points, MixtureDistribution[weights, #]& /@ Transpose[perSample]]
];

(* ---- the native batched sampler (gphip_nested_sampling): nestedSamplingInternal's job (BayesianStatistics.wl:859-1040)
   done inside the library, `Walkers` constrained-prior chains in lock step = one batched likelihood call per Metropolis
   step instead of "MonteCarloSteps" sequential calls of the closure.  The rows come back in generation order; they are
   wrapped into the reference's "Samples" association (:903-913, :1005-1013) and post-processed by the reference's OWN
   evidenceSampling (:1158-1291), so the returned object has exactly the keys nestedSampling gives. ---- *)
Options[nestedSamplingHIP] = Join[Options[nestedSampling], {"Walkers" -> 32, "Seed" -> 0, "PriorKinds" -> Automatic}];

uniformPriorQ[prior_, p_] := MatchQ[prior, "Uniform" | _UniformDistribution |
	ProductDistribution[(_UniformDistribution | {_UniformDistribution, _Integer})..]];

(* Any SEPARABLE prior (a ProductDistribution of univariate distributions -- or one univariate distribution for one
   parameter): its "LogPriorPDFFunction" (BayesianStatistics.wl:256-274) is the sum of the factors' log densities, which travel
   to the native driver as tables on a uniform grid over each parameter's {min, max}; zeros of a density become -1.*^300 (read
   as -Infinity by the shim).  $Failed for anything else (joint distributions): those runs stay with nestedSampling. *)
priorFactors[ProductDistribution[d__], p_] := With[{f = Flatten[Replace[{d}, {dist_, n_Integer} :> ConstantArray[dist, n], {1}]]},
	If[Length[f] === p && AllTrue[f, UnivariateDistributionQ], f, $Failed]];
priorFactors[d_?UnivariateDistributionQ, 1] := {d};
priorFactors[__] := $Failed;
priorTables[factors_List, params_, m_Integer] := MapThread[
	Function[{dist, spec},
		Clip[Replace[N @ Log @ PDF[dist, N @ Subdivide[spec[[2]], spec[[3]], m - 1]], Except[_Real] -> -1.*^300, {1}], {-1.*^300, 1.*^300}]],
	{factors, params}];
(* the starting pool: draws from the prior that fall inside the parameter ranges (generateStartingPoints, BS:1046-1068) *)
priorPool[prior_, params_, pool_Integer] := Module[{pts = {}, draw, tries = 0},
	While[Length[pts] < pool && tries++ < 50,
		draw = RandomVariate[prior, 2 pool];
		If[VectorQ[draw], draw = List /@ draw];
		pts = Join[pts, Select[draw, And @@ Thread[params[[All, 2]] <= # <= params[[All, 3]]]&]]
	];
	If[Length[pts] >= pool, N @ pts[[;; pool]], $Failed]
];

nestedSamplingHIP[inferenceObject[assoc_?AssociationQ], opts : OptionsPattern[]] /;
	KeyExistsQ[Lookup[assoc, "GaussianProcessData", <||>], "HIPHandle"] := Module[{
	h = assoc["GaussianProcessData", "HIPHandle"],
	params = assoc["Parameters"], p, pool, kinds, factors, nsOpts, start, rows, samples, result,
	pwFlags = Lookup[assoc["GaussianProcessData"], "PointwiseFunctions", {False, False}]
},
	p = Length[params];
	kinds = Replace[OptionValue["PriorKinds"], Automatic :> If[uniformPriorQ[assoc["PriorDistribution"], p], ConstantArray[0, p], $Failed]];
	factors = If[kinds === $Failed, priorFactors[assoc["PriorDistribution"], p], $Failed];
	(* point-dependent nugget / mean functions live in this kernel process, and so do NON-separable priors: those runs go
	   through the reference's own driver with the GPU closure *)
	If[ (kinds === $Failed && factors === $Failed) || Or @@ pwFlags,
		Return @ nestedSampling[inferenceObject[assoc], Sequence @@ FilterRules[{opts}, Options[nestedSampling]]]
	];
	start = Replace[OptionValue["StartingPoints"], Except[_?(MatrixQ[#, NumericQ]&)] :> Lookup[assoc, "StartingPoints", {}]];
	pool = If[MatrixQ[start], Length[start], OptionValue["SamplePoolSize"]];
	touch[h];
	nsOpts = N @ {pool, OptionValue["MaxIterations"], OptionValue["MinIterations"], OptionValue["MonteCarloSteps"], OptionValue["Walkers"],
		OptionValue["TerminationFraction"], Sequence @@ OptionValue["MinMaxAcceptanceRate"], OptionValue["Seed"]};
	rows = If[ kinds =!= $Failed,
		gpNested[h, N @ params[[All, {2, 3}]], kinds, nsOpts, If[MatrixQ[start], N @ start, {}]],
		(* separable prior: tabulated factors, pool drawn here from the prior itself *)
		If[!MatrixQ[start], start = priorPool[assoc["PriorDistribution"], params, pool]];
		If[ start === $Failed,
			Return @ nestedSampling[inferenceObject[assoc], Sequence @@ FilterRules[{opts}, Options[nestedSampling]]]
		];
		gpNestedTab[h, N @ params[[All, {2, 3}]], priorTables[factors, params, 2049], nsOpts, N @ start]
	];
	If[ !MatrixQ[rows], Return["Bad likelihood function"]];     (* BayesianStatistics.wl:917-921 *)
	samples = Association @ MapIndexed[
		Function[{row, index},
			First[index] -> <|
				"Point" -> row[[;; p]],
				"LogLikelihood" -> row[[p + 1]],
				"LogPriorPDF" -> row[[p + 2]],
				"AcceptanceRate" -> If[First[index] <= pool, Missing["InitialSample"], row[[p + 3]]]
			|>
		],
		rows
	];
	result = evidenceSampling[
		<|
			"Samples" -> samples,
			"SamplePoolSize" -> pool,
			"GeneratedNestedSamples" -> Length[rows] - pool,
			"TotalSamples" -> Length[rows],
			"ParameterRanges" -> CoordinateBounds[rows[[All, ;; p]]]
		|>,
		params[[All, 1]],
		Sequence @@ FilterRules[{opts}, Options[evidenceSampling]]
	];
	If[ TrueQ @ AssociationQ[result],
		inferenceObject[Join[assoc, <|"StartingPoints" -> rows[[;; pool, ;; p]]|>, result]],
		result
	]
];
nestedSamplingHIP[obj_, opts___] := nestedSampling[obj, Sequence @@ FilterRules[{opts}, Options[nestedSampling]]];

(* The reference's own definition (BayesianGaussianProcess.wl:343-346) matches a HIP object just as well -- its LHS
   differs from ours only inside a PatternTest, which WL cannot order by specificity -- and it was defined first.
   So the HIP rule is PREPENDED to the down-values instead of appended by an ordinary definition.  (With the
   reference-shaped "ModelFunctions" above, the reference's rule would still work on a HIP object: S interpreted
   kernel builds + LU factorisations instead of one batched GPU pass.) *)
Unprotect[predictFromGaussianProcess];
DownValues[predictFromGaussianProcess] = Prepend[
	DownValues[predictFromGaussianProcess],
	HoldPattern[predictFromGaussianProcess[inferenceObject[result_?hipObjectQ], pts_List]] :> hipPredict[result, pts]
];

(* predictiveDistribution (BayesianStatistics.wl:1373-1387) needs a "GeneratingDistribution", which a GP object does
   not carry; for HIP-backed GP objects it forwards to the prediction above.  The "MaximumLikelihood" / "MAP"
   forms (:1389-1416) reduce "Samples" to one element and re-enter here. *)
bestSample[result_, f_] := Append[result, "Samples" -> TakeLargestBy[result["Samples"], f, 1]];
Unprotect[predictiveDistribution];
DownValues[predictiveDistribution] = Join[
	{
		HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List]] :>
			hipPredict[result, pts],
		HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List, "MaximumLikelihood"]] :>
			hipPredict[bestSample[result, #LogLikelihood &], pts],
		HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List, "MAP"]] :>
			hipPredict[bestSample[result, #LogLikelihood + #LogPriorPDF &], pts]
	},
	DownValues[predictiveDistribution]
];

End[]
EndPackage[]
